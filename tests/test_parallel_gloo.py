"""CPU, world_size 2, gloo: the data-parallel step (bucketed gradient sum overlapped with backward + global token
normaliser) equals the single-process full-batch step.  The arithmetic stand-in is the CPU oracle (tests may use it);
bmt_amd.parallel itself is backend-agnostic torch.distributed plumbing and runs unchanged over RCCL."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from bmt_amd import synthetic as syn
from oracle import bmt_oracle as orc

V, B, TV, TA, TC = 11, 4, 9, 14, 7


class OracleCaptioner(torch.nn.Module):
    """nn.Module shell around the functional oracle so hooks / optimizers have Parameters to work with."""

    def __init__(self, cfg):
        super().__init__()
        sd = orc.init_captioning_params(cfg, V, seed=0, glove=None)
        self.names = list(sd.keys())
        self.plist = torch.nn.ParameterList([torch.nn.Parameter(sd[k].clone()) for k in self.names])
        self.cfg = cfg

    def loss_sum(self, fs, caps):
        p = dict(zip(self.names, self.plist))
        x, y = caps[:, :-1], caps[:, 1:]
        masks = orc.make_masks(fs, x, syn.PAD_IDX)
        pred = orc.bimodal_transformer(p, self.cfg, fs, x, masks)
        return orc.label_smoothing_kl(pred, y, self.cfg.smoothing, syn.PAD_IDX), (y != syn.PAD_IDX).sum()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_path, bucket_bytes, collective="allreduce"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bmt_amd.parallel import GradientReducer, global_sum
    cfg = syn.cfg_tiny(dout_p=0.0)
    model = OracleCaptioner(cfg)
    batch = syn.make_cap_batch(cfg, B, TV, TA, TC, V, seed=77)
    lo, hi = rank * B // world, (rank + 1) * B // world
    fs = {k: v[lo:hi] for k, v in batch["feature_stacks"].items()}
    caps = batch["captions"][lo:hi]
    red = GradientReducer(model.parameters(), bucket_bytes=bucket_bytes, collective=collective)
    assert len(red.buckets) > 1
    for it in range(2):          # second iteration checks zero_grad / re-arming of the buckets
        red.zero_grad()
        kl, ntok = model.loss_sum(fs, caps)
        loss = kl / global_sum(ntok)
        loss.backward()
        red.finish()
    if rank == 0:
        torch.save({k: p.grad.clone() for k, p in zip(model.names, model.plist)}, out_path)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("bucket_bytes,collective", [(64 << 10, "allreduce"), (256 << 10, "allreduce"), (64 << 10, "rs_ag")])
def test_two_rank_gradient_sum_equals_full_batch(tmp_path, bucket_bytes, collective):
    """(collective "rs_ag": every bucket as reduce-scatter + all-gather -- SURVEY.md section 5's form for the point-to-point xGMI mesh --
    leaves the same sums in the same flat buffers)"""
    out = str(tmp_path / "grads.pt")
    mp.spawn(_worker, args=(2, _free_port(), out, bucket_bytes, collective), nprocs=2, join=True)
    got = torch.load(out)
    cfg = syn.cfg_tiny(dout_p=0.0)
    model = OracleCaptioner(cfg)
    batch = syn.make_cap_batch(cfg, B, TV, TA, TC, V, seed=77)
    kl, ntok = model.loss_sum(batch["feature_stacks"], batch["captions"])
    (kl / ntok).backward()
    for k, p in zip(model.names, model.plist):
        torch.testing.assert_close(got[k], p.grad, rtol=1e-4, atol=1e-6, msg=k)


def test_reducer_single_process_is_a_noop_binding():
    from bmt_amd.parallel import GradientReducer
    lin = torch.nn.Linear(4, 3)
    red = GradientReducer(lin.parameters())
    lin(torch.ones(2, 4)).sum().backward()
    red.finish()
    bi, si = red._slot[lin.weight]
    assert lin.weight.grad.data_ptr() == red.buckets[bi]["views"][si].data_ptr()
    torch.testing.assert_close(lin.weight.grad, torch.full((3, 4), 2.0))
    red.zero_grad()
    assert float(lin.weight.grad.abs().sum()) == 0.0


def test_reducer_lays_grouped_gradients_back_to_back():
    """GradientReducer(groups=...): the gradients of a group are adjacent, in group order, inside one bucket (what lets the
    fused Q/K/V weight gradient be written by one GEMM), whatever the registration order and the bucket size"""
    import torch
    from bmt_amd.parallel import GradientReducer
    ps = [torch.nn.Parameter(torch.randn(n, 3)) for n in (4, 5, 6, 7, 8, 9)]
    groups = [[ps[4], ps[1], ps[2]]]
    red = GradientReducer(ps, bucket_bytes=40 * 4, groups=groups)
    try:
        a, b, c = ps[4].grad, ps[1].grad, ps[2].grad
        assert a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr() == c.untyped_storage().data_ptr()
        assert b.storage_offset() == a.storage_offset() + a.numel() and c.storage_offset() == b.storage_offset() + b.numel()
        # every parameter still owns a distinct slot and accumulates into it
        seen = set()
        for p in ps:
            key = (p.grad.untyped_storage().data_ptr(), p.grad.storage_offset())
            assert key not in seen
            seen.add(key)
        (sum((p * (i + 1)).sum() for i, p in enumerate(ps))).backward()
        for i, p in enumerate(ps):
            assert torch.equal(p.grad, torch.full_like(p, float(i + 1)))
    finally:
        red.remove()


# ---------------------------------------------------------------- train_prop under data parallelism
def _prop_cfg():
    cfg = syn.cfg_tiny(procedure="train_prop", dout_p=0.0)
    cfg.anchors_num_audio, cfg.anchors_num_video = 3, 5
    cfg.conv_layers_audio, cfg.conv_layers_video = [16, 16], [16, 16]
    cfg.kernel_sizes = {"audio": [1, 5], "video": [3, 7]}
    return cfg


def _prop_params(golden_name="tiny_prop.npz"):
    import numpy as np
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", golden_name))
    sd = {k[3:]: torch.from_numpy(np.array(z[k])) for k in z.files if k.startswith("sd/")}
    anchors = {"audio": [float(a) for a in z["anchors_audio"]], "video": [float(a) for a in z["anchors_video"]]}
    return sd, anchors


def _prop_loss(p, cfg, anchors, fs, targets, count_reduce=None):
    masks = orc.make_masks(fs, None, syn.PAD_IDX)
    _, loss, _, _ = orc.multimodal_proposal_generator(p, cfg, anchors, fs, targets, masks, count_reduce=count_reduce)
    return loss


def _prop_worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bmt_amd.parallel import GradientReducer, global_sum, shard_proposal_batch
    cfg = _prop_cfg()
    sd, anchors = _prop_params()
    names = list(sd.keys())
    plist = [torch.nn.Parameter(sd[k].clone()) for k in names]
    batch = syn.make_prop_batch(cfg, 4, 9, 14, seed=9, events_per_video=2)
    fs, tg = shard_proposal_batch(batch["feature_stacks"], batch["targets"], rank, world)
    red = GradientReducer(plist, bucket_bytes=64 << 10)
    red.zero_grad()
    loss = _prop_loss(dict(zip(names, plist)), cfg, anchors, fs, tg, count_reduce=global_sum)
    loss.backward()
    red.finish()
    total = global_sum(loss.detach())
    if rank == 0:
        torch.save({"grads": {k: p.grad.clone() for k, p in zip(names, plist)}, "loss": total}, out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_proposal_step_equals_full_batch(tmp_path):
    """per-rank LOCAL sums over GLOBAL obj / noobj counts + a gradient SUM == the full-batch MSE / BCE means of
    model/proposal_generator.py:316-321 (ranks hold different numbers of events: 2+3 vs 2+3 videos' events re-based)"""
    out = str(tmp_path / "prop.pt")
    mp.spawn(_prop_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    cfg = _prop_cfg()
    sd, anchors = _prop_params()
    p = {k: v.clone().requires_grad_() for k, v in sd.items()}
    batch = syn.make_prop_batch(cfg, 4, 9, 14, seed=9, events_per_video=2)
    loss = _prop_loss(p, cfg, anchors, batch["feature_stacks"], batch["targets"])
    loss.backward()
    torch.testing.assert_close(got["loss"], loss.detach(), rtol=1e-5, atol=1e-5)
    for k in p:
        torch.testing.assert_close(got["grads"][k], p[k].grad, rtol=2e-4, atol=2e-6, msg=k)


def test_shard_proposal_batch_rebases_targets():
    from bmt_amd.parallel import shard_proposal_batch
    cfg = _prop_cfg()
    batch = syn.make_prop_batch(cfg, 4, 9, 14, seed=9, events_per_video=2)
    seen = 0
    for r in range(2):
        fs, tg = shard_proposal_batch(batch["feature_stacks"], batch["targets"], r, 2)
        assert fs["rgb"].shape[0] == 2 and set(tg[:, 0].tolist()) <= {0.0, 1.0}
        seen += tg.shape[0]
    assert seen == batch["targets"].shape[0]


def test_buckets_follow_the_flush_stages():
    """round 6: a gradient bucket holds parameters of ONE flush stage (generator + decoder | encoder layer k), so that its all-reduce starts at
    that stage's flush point instead of waiting for a later layer's; bucketize / bucket_flush_map are pure arithmetic over the parameters"""
    import torch.nn as nn
    from bmt_amd import parallel

    class Enc(nn.Module):
        def __init__(self):
            super().__init__()
            self.layers = nn.ModuleList([nn.Linear(64, 64) for _ in range(3)])

    class M(nn.Module):
        def __init__(self):
            super().__init__()
            self.encoder = nn.Module()
            self.encoder.encoder_AV = Enc()
            self.decoder = nn.Linear(64, 32)
            self.generator = nn.Linear(32, 16)
    m = M()
    stage, points = parallel.flush_stages(m)
    assert len(points) == 4 and stage[id(m.generator.weight)] == 0 and stage[id(m.encoder.encoder_AV.layers[2].weight)] == 1 \
        and stage[id(m.encoder.encoder_AV.layers[0].bias)] == 3
    big = 1 << 30
    plain = parallel.bucketize(list(m.parameters()), big)
    assert len(plain) == 1                                     # everything fits one bucket ...
    aligned = parallel.bucketize(list(m.parameters()), big, None, stage)
    assert len(aligned) == 4                                   # ... but a bucket never mixes stages
    for ps in aligned:
        assert len({stage[id(p)] for p in ps}) == 1
    assert [id(p) for ps in aligned for p in ps] == [id(p) for p in reversed(list(m.parameters()))]      # order unchanged
    fm = parallel.bucket_flush_map(m, big)
    assert fm["bytes_final_at_point"] == [4 * (64 * 32 + 32 + 32 * 16 + 16)] + [4 * (64 * 64 + 64)] * 3
    fm0 = parallel.bucket_flush_map(m, big, aligned=False)
    assert fm0["bytes_final_at_point"][-1] == fm0["total_bytes"]      # one bucket: everything waits for the end
    # the reducer's buckets are the same lists
    red = parallel.GradientReducer(list(m.parameters()), bucket_bytes=big, stage_of=stage)
    assert [[id(p) for p in b["params"]] for b in red.buckets] == [[id(p) for p in ps] for ps in aligned]
    red.remove()
