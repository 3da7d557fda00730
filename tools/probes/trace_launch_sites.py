"""every library call of one eager train_cap step (configs[1]) by the Python site that issued it: API name, count, call chain inside bmt_amd --
to read the step's launch list as a list of decisions.   usage (GPU box): PYTHONPATH=. python tools/probes/trace_launch_sites.py"""
import argparse, collections, traceback
import torch
import bench
from bmt_amd import ops

args = argparse.Namespace(batch=32, batches=1, dp_collective="auto")
dev = torch.device("cuda:0")
step, (fs, caps), _, _ = bench.build_cap(args, dev, 0, 1)
for _ in range(3):
    step(fs, caps)
torch.cuda.synchronize()
log = collections.OrderedDict()
real = ops.lib


class Proxy:
    def __getattr__(self, name):
        fn = getattr(real, name)
        if not name.startswith("bmt_") or name in ("bmt_last_error",):
            return fn

        def inner(*a, **kw):
            st = [f"{f.name}:{f.lineno}" for f in traceback.extract_stack()[:-1] if "bmt_amd" in f.filename][-5:]
            key = (name, " < ".join(reversed(st)))
            log[key] = log.get(key, 0) + 1
            return fn(*a, **kw)
        return inner


ops.lib = Proxy()
step(fs, caps)
torch.cuda.synchronize()
ops.lib = real
for (name, st), n in log.items():
    print(f"{n:3d} x {name:34s} {st}")
print(sum(log.values()), "library calls")
