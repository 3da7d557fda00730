"""-m gpu: the data-parallel launch modes on one GPU through a 1-rank RCCL group with forced collectives
(tools/dp_smoke_1gpu.py: hipGraph + eager all-reduce, eager with bucket all-reduces from the backward hooks, eager with the
all-reduce after the backward pass), each against the single-process step."""
import importlib.util
import os
import socket

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_data_parallel_modes_on_a_one_rank_rccl_group(capsys):
    import torch.distributed as dist
    if dist.is_initialized():
        pytest.skip("a process group is already initialised in this process")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    spec = importlib.util.spec_from_file_location("dp_smoke_1gpu", os.path.join(ROOT, "tools", "dp_smoke_1gpu.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    try:
        ok = mod.main()
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()
    out = capsys.readouterr().out
    assert ok and "DP-SMOKE OK" in out and "DIFFER" not in out, out[-2000:]
