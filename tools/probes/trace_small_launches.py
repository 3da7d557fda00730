"""who issues the small conversion / glue launches of one eager train_cap step (configs[1]): every call of ops.make_planes, ops.zero_, ops.add_,
ops.cat2, lib.bmt_add with the Python call chain that issued it -- to find launches a producer could have made unnecessary.
usage (GPU box): PYTHONPATH=. python tools/probes/trace_small_launches.py"""
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
log = collections.Counter()


def wrap(name, fn):
    def inner(*a, **kw):
        st = [f"{f.name}:{f.lineno}" for f in traceback.extract_stack()[:-1] if "bmt_amd" in f.filename][-4:]
        shp = tuple(a[0].shape) if a and hasattr(a[0], "shape") else ""
        log[(name, shp, " < ".join(reversed(st)))] += 1
        return fn(*a, **kw)
    return inner


ops.make_planes = wrap("make_planes", ops.make_planes)
ops.zero_ = wrap("zero_", ops.zero_)
ops.dropout_raw = wrap("dropout_raw", ops.dropout_raw)
step(fs, caps)
torch.cuda.synchronize()
for (name, shp, st), n in sorted(log.items(), key=lambda kv: (kv[0][0], -kv[1])):
    print(f"{n:3d} x {name:12s} {str(shp):22s} {st}")
