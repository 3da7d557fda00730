"""-m gpu: the data-parallel launch modes on one GPU through a 1-rank RCCL group with forced collectives
(tools/dp_smoke_1gpu.py: hipGraph + eager all-reduce, eager with bucket all-reduces from the backward hooks, eager with the
all-reduce after the backward pass, and ONE hipGraph with the bucket all-reduces captured inside it), each against the single-process step."""
import os
import socket

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_data_parallel_modes_on_a_one_rank_rccl_group():
    """runs tools/dp_smoke_1gpu.py in a child process: the c10d / RCCL watchdog threads of this image have been seen to throw from
    their destructors after a clean destroy_process_group (std::terminate -> a core dump that would take the whole pytest session
    with it); the child prints its verdict, flushes and leaves through os._exit."""
    import subprocess
    import sys
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for attempt in range(2):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dp_smoke_1gpu.py"), "--graph-overlap"], env=env, capture_output=True, text=True,
                           timeout=600)
        out = r.stdout
        if "DP-SMOKE" in out:
            break
        # (the child died before its verdict -- the image's c10d / RCCL start-up has been seen to abort now and then, 1 run in ~10 of the suite: one
        # more try with a fresh port; a child that PRINTED a verdict is never re-run)
        print(f"dp_smoke_1gpu.py left without a verdict (rc {r.returncode}): {r.stderr[-500:]}")
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            env["MASTER_PORT"] = str(s.getsockname()[1])
    assert "DP-SMOKE OK" in out and "DIFFER" not in out, (r.returncode, out[-2000:], r.stderr[-2000:])
    assert "dp_graph_overlap" in out, out[-2000:]
