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


def test_executed_figures_ride_beside_the_padded_ones():
    """round 6: every timed launch carries the FLOPs / bytes of the rows that exist next to the padded-dense ones; a launch that gives none
    counts its padded figures as executed (4-tuples of the earlier rounds still summarise)"""
    steps = [[("gemm", 1.0, 10.0, 4.0, 7.5, 3.0), ("attn", 2.0, 20.0, 2.0, 11.0, 1.5), ("gemm", 1.0, 10.0, 4.0)] for _ in range(5)]
    summ, used = bench.summarize_intervals(steps)
    assert used == 5
    assert summ["gemm"]["flops"] == 20.0 and summ["gemm"]["xflops"] == 17.5 and summ["gemm"]["xbytes"] == 7.0
    assert summ["attn"]["xflops"] == 11.0 and summ["attn"]["bytes"] == 2.0


def test_step_flops_over_the_rows_that_exist():
    """cap_step_flops: SURVEY.md 8d's padded-dense 3.257 TFLOP per 32 full-length samples; row-wise products scale with the rows, attention cores
    with Lq * Lk, the decoder's products against a memory with the memory's length; caption rows count in full"""
    full = bench.cap_step_flops([800] * 32, [256] * 32)
    assert abs(full / 3.257e12 - 1.0) < 1e-3
    half = bench.cap_step_flops([400] * 32, [128] * 32)
    assert 0.25 * full < half < 0.5 * full            # cores quarter, row-wise products halve, the decoder's own products stay
    one = bench.cap_step_flops([800], [256])
    assert abs(one * 32 - full) < 1e-6 * full
    # an all-audio change moves only what depends on the audio length
    assert bench.cap_step_flops([400], [256]) < one and bench.cap_step_flops([800], [128]) < one


def test_timer_executed_rows_and_attention_extents():
    t = bench.KernelTimer()
    t.set_valid({25600: 19508, 8192: 6242}, {800: [800, 400], 256: [256, 128]})
    assert t.xrows(25600, True) == 19508 and t.xrows(25600, False) == 25600 and t.xrows(960, True) == 960
    qk, q, k = t.xattn(2, 800, 256, True, True)
    assert qk == 800 * 256 + 400 * 128 and q == 1200 and k == 384
    qk, q, k = t.xattn(2, 800, 256, True, False)          # padded key side: counts in full
    assert qk == 1200 * 256 and k == 512
