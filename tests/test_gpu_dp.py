"""-m gpu: the data-parallel launch modes on one GPU through a 1-rank RCCL group with forced collectives (tools/dp_smoke_1gpu.py,
run in a child process so that the process group does not leak into the other tests)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_data_parallel_modes_on_a_one_rank_rccl_group():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dp_smoke_1gpu.py")], capture_output=True, text=True, env=env,
                       timeout=600)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0 and "DP-SMOKE OK" in r.stdout, tail
    for mode in ("dp_graph", "dp_eager_overlap", "dp_eager_after"):
        assert f"{mode}" in r.stdout and "DIFFER" not in r.stdout, tail
