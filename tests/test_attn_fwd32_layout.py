"""Lane algebra of attn_fwd32_kernel (bmt_amd/csrc/attention_bf16.hip) on the CPU: the LDS image the DMA builds with its source-side
swizzles, the row-fragment and transposing reads, and the 32x32x16 MFMA operand / result layouts reproduce K.Q^T and V^T.P^T exactly,
with no LDS bank conflict under the documented bank model (tools/probes/attn_fwd32_layout.py is the emulator; the GPU parity of the
kernel itself is tests/test_gpu_kernels.py::test_attention_forward_32_query_kernel)."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _emulator(name="attn_fwd32_layout"):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools", "probes"))      # the backward emulator imports the forward one
    try:
        spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", "probes", name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        sys.path.pop(0)
    return mod


@pytest.mark.parametrize("dk", [256, 128])
def test_lane_algebra_and_banks(dk):
    assert _emulator().check(dk)


@pytest.mark.parametrize("dk", [256, 128])
def test_split_backward_dkv_lane_algebra_and_banks(dk):
    """attn_bwd_dkvg8_kernel / attn_bwd_dkvg_kernel (the split backward's dK / dV products): both MFMA operands are transposing reads of
    images whose row is the reduction index q -- the q' / dO tiles in the dual-purpose swizzle, the P / dS column blocks with
    chunk ^ ((row & 3) << 2) -- dV^T = dO^T . P against numpy for all four key groups, without bank conflicts"""
    assert _emulator("attn_bwd_split_layout").check(dk)


def test_split_backward_emission_layout():
    """the dQ kernel's P / dS emission: after the v_permlane32_swap exchange every lane holds 8 consecutive keys of its query row"""
    assert _emulator("attn_bwd_split_layout").check_emit()
