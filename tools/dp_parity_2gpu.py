#!/usr/bin/env python
"""Data-parallel PARITY on two (or more) GPUs: `python -m torch.distributed.run --nproc-per-node 2 tools/dp_parity_2gpu.py`.

Every rank takes its contiguous shard of ONE global batch and runs the HIP CaptioningTrainStep over RCCL in each launch mode bench.py can
pick (eager with bucket all-reduces from the backward pass, two hipGraphs with the all-reduce between them, one hipGraph with the
collectives captured inside it) and with both bucket collectives (all-reduce, reduce-scatter + all-gather).  The result is compared with
the ORACLE's full-batch step on the CPU (oracle/bmt_oracle.py -- the checker, test infrastructure): the global loss and, per parameter,
gradient-sum / global token count against the gradient of the full-batch loss.  Learning rate 0 and dropout 0: the weights never move, so
every step of every mode has the same answer and a captured graph's warm-up steps do not matter.
Rank 0 prints one line per arm and `DP-PARITY OK` / `DP-PARITY FAILED`."""
import contextlib
import io
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bmt_amd import ops, synthetic as syn  # noqa: E402
from bmt_amd.model.captioning_module import BiModalTransformer  # noqa: E402
from bmt_amd.train import CaptioningTrainStep  # noqa: E402

PER_TENSOR, GLOBAL, LOSS_TOL = 0.05, 0.01, 1e-3      # the bars of tests/test_gpu_model.py::_check_grads and of the log-prob parity


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    V, Tv, Ta, Tc, Bg = 1000, 64, 200, 12, 4 * world
    cfg = syn.cfg_config0(dout_p=0.0)
    cfg.device, cfg.lr = str(dev), 0.0
    batch = syn.make_cap_batch(cfg, Bg, Tv, Ta, Tc, V, seed=5)
    lo, hi = rank * Bg // world, (rank + 1) * Bg // world
    fs = {k: v[lo:hi].to(dev) for k, v in batch["feature_stacks"].items()}
    caps = batch["captions"][lo:hi].to(dev)
    site0 = ops._site_counter[0]

    def build():
        ops._site_counter[0] = site0
        torch.manual_seed(0)
        with contextlib.redirect_stdout(io.StringIO()):
            return BiModalTransformer(cfg, syn.FakeTrainDataset(V, syn.make_glove(V, cfg.d_model_caps))).to(dev)

    # ---- the checker: the oracle's full-batch step on the CPU (every rank computes it; cheap at this size)
    from oracle import bmt_oracle as orc
    ref_model = build()
    sd = {k: v.detach().cpu().clone() for k, v in ref_model.state_dict().items()}
    trainable = {k for k, p in ref_model.named_parameters() if p.requires_grad}
    p = {k: v.clone().requires_grad_(k in trainable) for k, v in sd.items()}
    orc.set_dropout(0.0)
    oloss, _, ntok = orc.train_cap_loss(p, cfg, batch["feature_stacks"], batch["captions"], syn.PAD_IDX, cfg.smoothing)
    oloss.backward()
    ref = {k: p[k].grad.double() for k in trainable if p[k].grad is not None}
    nmax = max(float(g.norm()) for g in ref.values())
    del ref_model

    arms = [("eager+overlap", "allreduce"), ("hipgraph", "allreduce"), ("hipgraph+captured-allreduce", "allreduce"),
            ("eager+overlap", "rs_ag"), ("hipgraph+captured-allreduce", "rs_ag")]
    if "--no-graph-overlap" in sys.argv:
        arms = [a for a in arms if a[0] != "hipgraph+captured-allreduce"]
    all_ok = True
    for mode, coll in arms:
        model = build()
        step = CaptioningTrainStep(model, cfg, syn.PAD_IDX, data_parallel=True, static_grads=True, overlap=True, seed=77, bucket_bytes=1 << 20,
                                   collective=coll)
        if mode == "eager+overlap":
            for _ in range(2):
                loss, n = step(fs, caps)
        else:
            step.capture(fs, caps, warmup=1, collectives=(mode == "hipgraph+captured-allreduce"))
            for _ in range(2):
                loss, n = step.replay()
        torch.cuda.synchronize()
        scale = float(step.grad_scale)
        bad, e2, r2, worst = [], 0.0, 0.0, (0.0, "")
        if abs(float(loss) - float(oloss)) > LOSS_TOL:
            bad.append(f"loss {float(loss):.6f} vs oracle {float(oloss):.6f}")
        if int(n) != int(ntok):
            bad.append(f"global token count {int(n)} vs {int(ntok)}")
        for k, q in model.named_parameters():
            if k not in ref:
                continue
            got = q.grad.detach().double().cpu() * scale
            e, r = float((got - ref[k]).norm()), float(ref[k].norm())
            if r < 1e-4 * nmax:
                continue
            e2, r2 = e2 + e * e, r2 + r * r
            worst = max(worst, (e / r, k))
            if e > PER_TENSOR * r:
                bad.append(f"{k}: {e / r:.1%}")
        g = (e2 / max(r2, 1e-300)) ** 0.5
        if g > GLOBAL:
            bad.append(f"global gradient error {g:.2%}")
        flag = torch.tensor([0 if bad else 1], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        all_ok = all_ok and bool(int(flag[0]))
        if rank == 0:
            print(f"{mode:30s} {coll:10s} loss {float(loss):.6f} (oracle {float(oloss):.6f})  gradients: global {g:.3%}, worst tensor {worst[0]:.2%} "
                  f"({worst[1]})  {'ok' if not bad else 'DIFFER: ' + '; '.join(bad[:6])}", flush=True)
        del step, model
    dist.barrier()
    if rank == 0:
        print("DP-PARITY OK" if all_ok else "DP-PARITY FAILED", flush=True)
    sys.stdout.flush()
    os._exit(0)          # (destroy_process_group aborts now and then on this image: leave without the interpreter's teardown)


if __name__ == "__main__":
    main()
