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


def _worker(rank, world, port, out_path, bucket_bytes):
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
    red = GradientReducer(model.parameters(), bucket_bytes=bucket_bytes)
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


@pytest.mark.parametrize("bucket_bytes", [64 << 10, 256 << 10])
def test_two_rank_gradient_sum_equals_full_batch(tmp_path, bucket_bytes):
    out = str(tmp_path / "grads.pt")
    mp.spawn(_worker, args=(2, _free_port(), out, bucket_bytes), nprocs=2, join=True)
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
