"""The oracle is test infrastructure: nothing under bmt_amd/ (nor bench.py's timed path) may import or call it."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_product_never_touches_the_oracle_or_the_reference():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "bmt_amd")):
        for f in files:
            if not f.endswith((".py", ".hip", ".h", ".cpp", ".sh")):
                continue
            text = open(os.path.join(dirpath, f), errors="replace").read()
            if re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M) or "bmt_oracle" in text:
                bad.append(os.path.join(dirpath, f) + ": imports the oracle")
            if "/root/reference" in text:
                bad.append(os.path.join(dirpath, f) + ": reads /root/reference")
    assert not bad, "\n".join(bad)


def test_ops_reject_cpu_tensors():
    """no CPU fallback: a CPU tensor reaching an op is an error, not a silent eager path."""
    import pytest
    import torch
    from bmt_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.LayerNormFn.apply(torch.randn(2, 8), torch.ones(8), torch.zeros(8), 1e-5)
