"""-m gpu, needs >= 2 GPUs (skipped on the 1-GPU boxes of the build pool): the HIP train step on TWO ranks over RCCL -- every launch mode
bench.py can choose and both bucket collectives -- against the ORACLE's full-batch step (tools/dp_parity_2gpu.py): the first multi-GPU
box this suite meets produces parity evidence for the data-parallel path, not just a throughput number."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_two_rank_rccl_step_equals_the_oracles_full_batch_step():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), OMP_NUM_THREADS="8")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tools", "dp_parity_2gpu.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert "DP-PARITY OK" in r.stdout and "DIFFER" not in r.stdout, (r.returncode, r.stdout[-3000:], r.stderr[-2000:])
