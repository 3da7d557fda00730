"""bench.py's kernel timer arithmetic and its sanity gates (no GPU): per launch the median over the eager steps, summed per class;
a pass whose numbers contradict each other is reported as invalid instead of printed as a roofline (a host stall
inside one event interval put 13.5 ms of "attention backward" into an 8.7 ms step)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def _step(gemm=1.0, attn=2.0, extra=None):
    st = [("gemm", gemm, 10.0, 1.0), ("attn_bwd_enc", attn, 20.0, 2.0), ("gemm", gemm / 2, 5.0, 0.5)]
    return st + (extra or [])


def test_one_stalled_launch_does_not_reach_the_class_time():
    steps = [_step() for _ in range(6)] + [_step(attn=90.0)]            # one hipMalloc landed between an event and its kernel
    summ, used = bench.summarize_intervals(steps)
    assert used == 7
    assert abs(summ["attn_bwd_enc"]["ms"] - 2.0) < 1e-9                   # median per launch
    assert summ["attn_bwd_enc"]["ms_sum_of_means"] > 14.0                 # what the sum of raw intervals would have said
    assert summ["attn_bwd_enc"]["ms_max_step"] == 90.0
    assert summ["gemm"]["launches"] == 2 and abs(summ["gemm"]["ms"] - 1.5) < 1e-9
    assert summ["gemm"]["flops"] == 15.0 and summ["gemm"]["bytes"] == 1.5   # per step, not per pass


def test_steps_with_another_launch_sequence_are_left_out():
    steps = [_step(), _step(), _step(extra=[("gemm", 1.0, 1.0, 1.0)]), _step()]
    summ, used = bench.summarize_intervals(steps)
    assert used == 3 and summ["gemm"]["launches"] == 2
    assert bench.summarize_intervals([]) == ({}, 0)


def test_gates():
    ok = bench.roofline_gates({"a": 2.0, "b": 3.0}, eager_ms=10.0, ms_per_step=8.7)
    assert ok == []
    # round 3's driver line: classes 23.7 ms inside a 40.1 ms "eager step" next to an 8.74 ms timed step
    bad = bench.roofline_gates({"attn_bwd": 15.2, "gemm": 8.5}, eager_ms=40.1, ms_per_step=8.74)
    assert any("1.6 x" in b for b in bad) and any("attn_bwd alone" in b for b in bad)
    assert bench.roofline_gates({"a": 6.0, "b": 6.0}, eager_ms=10.0, ms_per_step=9.0)     # classes > the step they were timed in
    assert bench.roofline_gates({"a": 1.0}, eager_ms=None, ms_per_step=9.0) == ["no eager step time"]
