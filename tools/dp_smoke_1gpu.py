#!/usr/bin/env python
"""The data-parallel code path on ONE GPU: a 1-rank RCCL process group with the gradient reducer forced to issue its
all-reduces (sum over one rank = identity), in the three launch modes bench.py can take under torch.distributed.run:
  * eager, bucket all-reduces launched from the backward hooks (overlap),
  * eager, all-reduce after the backward pass,
  * two hipGraphs with the eager all-reduce between them (what `bench.py --gpus N` runs),
  * (--graph-overlap) ONE hipGraph with the bucket all-reduces captured inside it (`bench.py --dp-mode graph-overlap`).
Each must reproduce the losses of the plain single-process step (to 1e-5 relative: the step's atomics make runs differ in
the last fp32 bits).  Exercises NCCL/RCCL initialisation on the box,
collectives on the flat gradient buckets from autograd's hook thread, stream ordering against the captured graphs."""
import contextlib
import io
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
from bmt_amd import ops, synthetic as syn  # noqa: E402
from bmt_amd.model.captioning_module import BiModalTransformer  # noqa: E402
from bmt_amd.train import CaptioningTrainStep  # noqa: E402

def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    V, Tv, Ta, Tc, B = 1000, 64, 200, 12, 8
    cfg = syn.cfg_config0(dout_p=0.1)
    cfg.device = str(dev)
    batch = syn.make_cap_batch(cfg, B, Tv, Ta, Tc, V, seed=5)
    fs = {k: v.to(dev) for k, v in batch["feature_stacks"].items()}
    caps = batch["captions"].to(dev)


    def run(mode):
        first_site = ops._site_counter[0] if not hasattr(run, "site") else run.site
        run.site = first_site
        ops._site_counter[0] = first_site
        torch.manual_seed(0)
        with contextlib.redirect_stdout(io.StringIO()):
            model = BiModalTransformer(cfg, syn.FakeTrainDataset(V, syn.make_glove(V, cfg.d_model_caps))).to(dev)
        dp = mode != "single"
        step = CaptioningTrainStep(model, cfg, syn.PAD_IDX, data_parallel=dp, static_grads=True, overlap=(mode in ("dp_eager_overlap", "dp_graph_overlap")),
                                   seed=77)          # dropout seed = 77 + rank on every arm (rank 0 here)
        calls = [0]
        if dp:
            step.reducer.world = 2            # one rank, but issue every collective
            raw = dist.all_reduce

            def counted(*a, **k):
                calls[0] += 1
                return raw(*a, **k)
            dist.all_reduce = counted
        losses = []
        try:
            if mode in ("single", "dp_graph", "dp_graph_overlap"):
                step.capture(fs, caps, warmup=1, collectives=(mode == "dp_graph_overlap"))
                for _ in range(4):
                    loss, _ = step.replay()
                    losses.append(float(loss))
            else:
                for _ in range(5):
                    loss, _ = step(fs, caps)
                    losses.append(float(loss))
                losses = losses[1:]
        finally:
            if dp:
                dist.all_reduce = raw
        torch.cuda.synchronize()
        return losses, calls[0]


    ref, _ = run("single")
    ok = True
    modes = ("dp_graph", "dp_eager_overlap", "dp_eager_after") + (("dp_graph_overlap",) if "--graph-overlap" in sys.argv else ())
    for mode in modes:
        got, n = run(mode)
        # 1e-4: a missing collective or a wrong normaliser is a >= 1e-2 difference; fp32 atomics (bias column sums) make the fourth step's
        # loss of ONE configuration wander by ~3e-6 from run to run
        same = all(abs(a - b) <= 1e-4 * abs(b) for a, b in zip(got, ref))
        ok &= same and n > 0
        print(f"{mode:18s} all_reduce calls {n:3d}  losses {'match' if same else 'DIFFER from'} the single-process step {got if not same else ''}")
    print("single            ", ref)
    torch.cuda.synchronize()
    print("DP-SMOKE", "OK" if ok else "FAILED", flush=True)
    # no destroy_process_group: on this image it aborts now and then (SIGABRT from a c10d / RCCL watchdog thread, 2 runs in 8 under a
    # piped stdout); the caller leaves through os._exit right after the verdict
    return ok


if __name__ == "__main__":
    ok_ = main()
    sys.stdout.flush()
    os._exit(0 if ok_ else 1)       # skip the interpreter teardown: c10d watchdog threads can throw after destroy_process_group
