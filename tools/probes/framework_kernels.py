#!/usr/bin/env python
"""which call sites of the product launch FRAMEWORK kernels (aten fill / copy / add / cat ...) in one eagerly issued step: every aten op
that reaches the device is tallied by its innermost bmt_amd frame.   usage: framework_kernels.py [train_prop|train_cap]"""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


class Tally(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.c = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        launches = any(k in name for k in ("fill", "zero", "copy", "add", "cat", "clone", "flip", "mul", "sum", "stack", "pad", "contiguous",
                                            "index", "select_scatter", "ones", "full", "sub", "div", "_to_copy", "where", "eq", "ne"))
        dev = any(isinstance(a, torch.Tensor) and a.is_cuda for a in list(args) + list((kwargs or {}).values()))
        out = func(*args, **(kwargs or {}))
        if isinstance(out, torch.Tensor) and out.is_cuda:
            dev = True
        if launches and dev and "view" not in name and "as_strided" not in name and "empty" not in name:
            site = "?"
            for fr in reversed(traceback.extract_stack()[:-1]):
                if "bmt_amd" in fr.filename or fr.filename.endswith("bench.py"):
                    site = f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}"
                    break
            if site == "?":
                site = " <- ".join(f"{os.path.basename(fr.filename)}:{fr.lineno}" for fr in traceback.extract_stack()[-5:-1])
            self.c[(name, site)] += 1
        return out


def main():
    proc = sys.argv[1] if len(sys.argv) > 1 else "train_prop"
    args = type("A", (), {"batch": 0, "dp_collective": "allreduce", "batches": 1})()
    dev = torch.device("cuda", 0)
    step, inputs, _, _ = (bench.build_cap if proc == "train_cap" else bench.build_prop)(args, dev, 0, 1)
    for _ in range(2):
        step(*inputs)
    torch.cuda.synchronize()
    # (backward nodes run on the calling thread: the dispatch mode sees them too)
    with torch.autograd.set_multithreading_enabled(False), Tally() as t:
        step(*inputs)
    torch.cuda.synchronize()
    tot = sum(t.c.values())
    print(f"{proc}: {tot} aten calls on device tensors in one eager step (not every one is a kernel: views excluded, scalar math included)")
    for (name, site), n in t.c.most_common(60):
        print(f"{n:5d}  {name:40s} {site}")


if __name__ == "__main__":
    main()
