"""-m gpu: the driver's invocation of bench.py, short -- the line must come out with a valid kernel-timer pass (the eager pass behind the timed
region wraps every op the step issues: a keyword one of them grew and the wrapper did not take made the default invocation raise while every
A/B script, which runs with other flags, kept working)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_default_invocation_prints_a_valid_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2", "--no-cpu-baseline"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["metric"] and line["n_gpus"] == 1 and line["steps"] == 4 and line["value"] > 0
    assert line["roofline"]["valid"] and 0 < line["roofline"]["frac"] < 1
    assert line["attention_roofline"]["valid"]
    kc = line["kernel_classes"]
    assert "gemm_planes_dw_grouped_bf16" in kc and "gemm_planes_memory_grad_grouped_bf16" in kc
    from bmt_amd import ops
    if ops.RAW_FUSED:
        assert any(k.startswith("raw_attn_fused_f16") for k in kc) and any(k.startswith("raw_attn_fused_bf16") for k in kc)
