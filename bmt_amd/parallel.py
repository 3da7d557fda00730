"""Data parallelism for the train_cap / train_prop step: one process per GPU, RCCL (torch.distributed backend "nccl")
sum-all-reduce of gradients over xGMI, bucketed in reverse-autograd order and overlapped with the backward pass.

Replaces the reference's single-process ``nn.DataParallel`` (scripts/train_captioning_module.py:61), which per step
broadcasts every parameter, scatters inputs, gathers the (B,Tc,V) log-probs to GPU 0, computes the loss there and
reduces gradients to GPU 0.  Here every rank holds a replica and its own optimizer state; the only
traffic is one gradient sum (~202 MB fp32 at config[1]) plus one scalar (the global token count the loss is
normalised by, epoch_loops/captioning_epoch_loops.py:134-135).

xGMI is a point-to-point mesh, so buckets are sized to keep all seven links busy rather than for an NVSwitch-style
ring: a few large buckets (default 32 MB) launched as soon as their gradients are final -- generator and decoder first,
the encoder layers (71 % of the bytes) while the earlier encoder layers are still in backward.

Nothing here computes: it is torch.distributed plumbing, so it is exercised on CPU with the gloo backend in
tests/test_parallel_gloo.py (world_size 2) and runs unchanged over RCCL on the GPU box."""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def bucketize(params, bucket_bytes: int, groups=None, stage_of=None) -> List[list]:
    """the parameters of each gradient bucket, in bucket order (pure: no buffer is allocated -- GradientReducer and bucket_flush_map share it).
    Reverse registration order ~ the order autograd finishes gradients (generator -> decoder -> encoder); ``groups``: lists of parameters
    whose gradients must sit back to back, in the given order, inside one bucket (the fused Q/K/V weight gradient is ONE GEMM writing one
    [3D, d_in] block: ops.group_static_grad).  ``stage_of`` (id(parameter) -> the flush point at which its gradient is final, round 6): a
    bucket never holds parameters of two stages -- one that straddled encoder layers 1 and 0 held 28 MB of layer-1 gradients back until
    the END of the pass (bucket_flush_map: 49 % of the bytes final only then; 36 % with stage-aligned buckets)."""
    params = [p for p in params if p.requires_grad]
    order = list(reversed(params))
    member = {}
    for g in (groups or []):
        g = [p for p in g if p.requires_grad]
        if len(g) > 1:
            for p in g:
                member[id(p)] = g
    units, placed = [], set()
    for p in order:
        if id(p) in placed:
            continue
        u = member.get(id(p), [p])
        units.append(u)
        placed.update(id(x) for x in u)
    cur, cur_bytes, out, cur_stage = [], 0, [], None
    for u in units:
        nbytes = sum(p.numel() for p in u) * 4
        st = max(stage_of.get(id(p), 0) for p in u) if stage_of else None
        if cur and (cur_bytes + nbytes > bucket_bytes or st != cur_stage):
            out.append(cur)
            cur, cur_bytes = [], 0
        cur.extend(u)
        cur_bytes += nbytes
        cur_stage = st
    if cur:
        out.append(cur)
    return out


def flush_stages(model, layer_prefix: str = "encoder.encoder_AV.layers."):
    """(id(parameter) -> index of the flush point at which its gradient is final, the flush points' descriptions) for a model whose encoder
    layers live under ``layer_prefix``: generator / decoder / embedders 0, encoder layer k: N - k (bucket_flush_map)"""
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    layers = sorted({int(n[len(layer_prefix):].split(".")[0]) for n, _ in named if n.startswith(layer_prefix)})
    n_layers = (max(layers) + 1) if layers else 0
    points = [f"gradient reaches encoder layer {k}'s output" for k in range(n_layers - 1, -1, -1)] + ["end of the backward pass"]
    stage = {}
    for n, p in named:
        stage[id(p)] = (n_layers - int(n[len(layer_prefix):].split(".")[0])) if n.startswith(layer_prefix) else 0
    return stage, points


def bucket_flush_map(model, bucket_bytes: int = 32 << 20, groups=None, layer_prefix: str = "encoder.encoder_AV.layers.", aligned: bool = True) -> dict:
    """which gradient bucket becomes final at which flush point of the overlapped backward pass, and how many bytes are left for the end.

    The weight-gradient products of a backward pass are queued and issued as grouped launches at FLUSH POINTS (CaptioningTrainStep.
    _install_flush_points): the moments the gradient arrives at the output of encoder layer k (k = N-1 .. 0) -- everything behind that
    layer (generator, decoder, layers > k) has been differentiated -- and the end of the pass.  A parameter's gradient is final at the first
    flush after its own backward: the generator's and the decoder's at the first point, encoder layer k's at the point that enters layer
    k-1, layer 0's at the end.  A BUCKET is final when its last parameter is, and its all-reduce starts then; what becomes final only at
    the end of the pass cannot overlap anything (exposed communication + imbalance <= 25 % of the step for >= 6x at 8 GPUs).
    Pure arithmetic over the parameter list (runs without a GPU: bench.py --dry-run reports it)."""
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    name_of = {id(p): n for n, p in named}
    # flush points in the order they happen: entering layer N-1, ..., entering layer 0, end of the pass
    stage, points = flush_stages(model, layer_prefix)
    buckets = []
    for i, ps in enumerate(bucketize([p for _, p in named], bucket_bytes, groups, stage if aligned else None)):
        at = max(stage[id(p)] for p in ps)
        buckets.append({"bucket": i, "bytes": 4 * sum(p.numel() for p in ps), "parameters": len(ps), "final_at": at,
                        "first": name_of[id(ps[0])], "last": name_of[id(ps[-1])]})
    total = sum(b["bytes"] for b in buckets)
    by_point = [sum(b["bytes"] for b in buckets if b["final_at"] == i) for i in range(len(points))]
    return {"bucket_bytes": bucket_bytes, "stage_aligned_buckets": bool(aligned), "flush_points": points, "buckets": buckets, "bytes_final_at_point": by_point, "total_bytes": total,
            "bytes_after_last_layer_flush": by_point[-1], "fraction_after_last_layer_flush": (by_point[-1] / total) if total else 0.0}


class GradientReducer:
    """Owns flat gradient buckets; ``p.grad`` of every trainable parameter is a view into one of them, so autograd
    accumulates straight into communication buffers and the optimizer reads the reduced values in place."""

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_bytes: int = 32 << 20,
                 process_group: Optional[dist.ProcessGroup] = None, overlap: bool = True, groups=None, collective: str = "allreduce",
                 stage_of=None):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.overlap = overlap
        # how a bucket's gradient sum is formed:
        #   "allreduce": one sum all-reduce per bucket (RCCL picks ring / tree and its channel count);
        #   "rs_ag":     reduce-scatter + all-gather -- every rank reduces 1/world of the bucket and hands its shard to every peer; on the
        #                fully connected xGMI mesh both halves are world-1 simultaneous point-to-point transfers of 1/world of the bucket,
        #                one per link, instead of a ring's world-1 dependent steps.
        # Both leave the same sum in the same flat buffer (tests/test_parallel_gloo.py); which is faster on 8 GPUs is the driver's run to say
        # (bench.py --dp-collective).
        if collective not in ("allreduce", "rs_ag"):
            raise ValueError(f"GradientReducer: unknown collective {collective!r}")
        self.collective = collective
        self._stream_ordered = dist.is_initialized() and dist.get_backend(process_group) == "nccl"
        params = [p for p in params if p.requires_grad]
        self.buckets: List[dict] = []
        groups_of_params = bucketize(params, bucket_bytes, groups, stage_of)
        # ONE allocation for every bucket of a device (the flat buffers are consecutive slices of it): zero_grad is one fill over the arena
        # instead of one per bucket (8 launches at the head of every step at config[1])
        quantum = max(1, self.world) * 8
        sizes = [(sum(p.numel() for p in ps) + quantum - 1) // quantum * quantum for ps in groups_of_params]
        self._arena = torch.zeros(sum(sizes), dtype=torch.float32, device=params[0].device) if params else None
        off = 0
        for ps, n in zip(groups_of_params, sizes):
            self._make_bucket(ps, self._arena[off:off + n])
            off += n
        self._slot = {}   # Parameter (hashed by identity) -> (bucket index, slot in bucket)
        for bi, b in enumerate(self.buckets):
            for si, p in enumerate(b["params"]):
                self._slot[p] = (bi, si)
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in params]
        for p in params:       # lets the fused weight-gradient path (ops.static_grad / ops.grad_done) find its buffer and report
            p._bmt_static_grad = True
            p._bmt_on_grad = self._on_grad
        self._handles = []
        self._second = []
        self.zero_grad()

    def _make_bucket(self, ps, flat):
        # (flat: this bucket's slice of the arena, padded to a multiple of the world size -- of 8 floats per rank, so that shards stay
        # 32-byte aligned -- for the reduce-scatter form: the pad belongs to no parameter and stays zero)
        if any(p.device != flat.device for p in ps):
            raise RuntimeError("GradientReducer: the parameters of one reducer live on one device")
        views, off = [], 0
        for p in ps:
            views.append(flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        shard = None
        if self.collective == "rs_ag" and self.world > 1:
            shard = torch.zeros(flat.numel() // self.world, dtype=torch.float32, device=flat.device)
        self.buckets.append({"params": ps, "flat": flat, "views": views, "pending": len(ps), "shard": shard})

    def set_collective(self, collective: str):
        """switch how a bucket's sum is formed (between steps; a captured hipGraph keeps what it was captured with)"""
        if collective not in ("allreduce", "rs_ag"):
            raise ValueError(f"GradientReducer: unknown collective {collective!r}")
        self.collective = collective
        if collective == "rs_ag" and self.world > 1:
            for b in self.buckets:
                if b["shard"] is None:
                    b["shard"] = torch.zeros(b["flat"].numel() // self.world, dtype=torch.float32, device=b["flat"].device)

    def _reduce_bucket(self, b):
        """launch the sum of one bucket over the ranks (asynchronously): the handles go to self._handles, and -- where two collectives of
        one bucket cannot simply be queued behind each other -- the second half to self._second"""
        if self.collective == "allreduce" or self.world == 1:
            self._handles.append(dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            return
        h = dist.reduce_scatter_tensor(b["shard"], b["flat"], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        if self._stream_ordered:        # RCCL: collectives of a group run in issue order on the communication stream
            self._handles.append(h)
            self._handles.append(dist.all_gather_into_tensor(b["flat"], b["shard"], group=self.group, async_op=True))
        else:                           # gloo (the CPU tests): asynchronous works may overtake each other -- gather once the scatter is done
            self._second.append((h, b))

    def zero_grad(self, defer: bool = False):
        """in-place zero of the flat buffers (one fill over the arena they are slices of); keeps p.grad bound to its bucket view.  ``defer``:
        the fill is handed to the pass (ops.defer_beside) and issued beside its forward instead of in front of it -- the caller joins it
        before the backward pass (ops.join_deferred)."""
        if self._arena is not None:
            if self._arena.is_cuda:      # one library launch (bmt_zero) over the whole arena
                from . import ops as _ops
                if defer:
                    arena = self._arena
                    _ops.defer_beside(lambda: _ops.zero_(arena))
                else:
                    _ops.zero_(self._arena)
            else:
                self._arena.zero_()
        for b in self.buckets:
            b["pending"] = len(b["params"])
            for p, v in zip(b["params"], b["views"]):
                if p.grad is not v:
                    p.grad = v
                p._bmt_uses = 0
        self._handles = []
        self._second = []
        if self._arena is not None and self._arena.is_cuda:      # a new step: nothing of the last one is left to settle
            from . import ops as _ops
            _ops.context().gen_handles.clear()

    def _on_grad(self, p):
        bi, si = self._slot[p]
        b = self.buckets[bi]
        v = b["views"][si]
        if p.grad.data_ptr() != v.data_ptr():
            # autograd replaced the tensor (grad was None): copy into the bucket and re-bind
            v.copy_(p.grad)
            p.grad = v
        b["pending"] -= 1
        if b["pending"] == 0 and self.world > 1 and self.overlap:
            self._reduce_bucket(b)

    def finish(self):
        """wait for the in-flight buckets (and reduce any bucket whose hooks did not all fire, e.g. unused parameters)"""
        if self.world > 1:
            for b in self.buckets:
                if b["pending"] != 0 or not self.overlap:
                    self._reduce_bucket(b)
            for h, b in self._second:
                h.wait()
                self._handles.append(dist.all_gather_into_tensor(b["flat"], b["shard"], group=self.group, async_op=True))
            for h in self._handles:
                h.wait()
        self._handles = []
        self._second = []

    def remove(self):
        for h in self._hooks:
            h.remove()
        for b in self.buckets:
            for p in b["params"]:
                p._bmt_static_grad = False
                p._bmt_on_grad = None


def global_sum(t: torch.Tensor, group=None) -> torch.Tensor:
    """sum of a small tensor over ranks (the loss normaliser: n_tokens, or the obj / noobj cell counts of train_prop)"""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        t = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def shard_proposal_batch(feature_stacks, targets: torch.Tensor, rank: int, world: int):
    """contiguous split of a train_prop batch over ranks: videos [lo, hi) of every feature stack and the rows of
    ``targets`` ((n_events, 4): [batch idx, center s, length s, meta idx], datasets/proposal_dataset.py:133-166) that belong to
    them, with the batch index re-based to the shard."""
    B = next(iter(feature_stacks.values())).shape[0]
    lo, hi = rank * B // world, (rank + 1) * B // world
    fs = {k: v[lo:hi] for k, v in feature_stacks.items()}
    keep = (targets[:, 0] >= lo) & (targets[:, 0] < hi)
    t = targets[keep].clone()
    t[:, 0] -= lo
    return fs, t
