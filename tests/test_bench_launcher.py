"""CPU: `python bench.py --gpus N` starts its own ranks (the driver's form) and the ranks rendezvous, time and reduce the way
the GPU run does -- exercised with --dry-run (gloo, no HIP work), plus the stale-PMC refusal of the roofline's traffic field."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=300, env=e)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


def test_self_launch_two_ranks_dry_run():
    r, out = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert out and out["dry_run"] and out["n_gpus"] == 2 and out["steps"] == 3
    # the bucket -> flush-point map of the overlapped backward pass (round 6): every byte of the 50.4 M gradients lands at exactly one flush
    # point, buckets never straddle two stages, and what is final only at the end of the pass is encoder layer 0 alone
    fm = out["dp_flush_map"]
    assert fm["stage_aligned_buckets"] and len(fm["flush_points"]) == 3 and sum(fm["bytes_final_at_point"]) == fm["total_bytes"] == 201772320
    assert all(b["final_at"] in (0, 1, 2) for b in fm["buckets"]) and sum(b["bytes"] for b in fm["buckets"]) == fm["total_bytes"]
    assert all(("layers.0." in b["first"]) == ("layers.0." in b["last"]) == (b["final_at"] == 2) for b in fm["buckets"] if "encoder" in b["first"])
    assert fm["bytes_after_last_layer_flush"] == fm["bytes_final_at_point"][-1] and 0.3 < fm["fraction_after_last_layer_flush"] < 0.4
    # max over ranks of the wall time (rank 1 sleeps 2 ms per step), sum over ranks of the units (100 + 200)
    assert out["ms_per_step"] >= 1.9 and abs(out["value"] * out["ms_per_step"] * 1e-3 - 300.0) < 1e-6
    # the evidence that N ranks took part travels in the line itself: what the collective library counts (a sum all-reduce of ones), every
    # rank's device, the exposed all-reduce time and the bucket collective (--dp-collective auto tries both forms)
    assert out["rccl_ranks_seen"] == 2
    assert sorted(d["rank"] for d in out["rank_devices"]) == [0, 1] and all("device" in d for d in out["rank_devices"])
    assert "allreduce_exposed_ms" in out and out["dp_collective"] in ("allreduce", "rs_ag")
    assert out["dp_collectives_tried"] == ["allreduce", "rs_ag"]
    r, out = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--dry-run", "--dp-collective", "rs_ag"])
    assert r.returncode == 0 and out["dp_collective"] == "rs_ag" and out["dp_collectives_tried"] == ["rs_ag"]


def test_captured_allreduce_trial_runs_in_child_processes_and_is_bounded():
    """--dp-mode auto (N > 1) tries the one-graph mode with the collectives captured inside it only after child processes -- one per rank,
    their own process group -- have shown that it works within a time bound; a child that hangs is killed and the mode is left out (the
    protocol over gloo: port broadcast, children's rendezvous, the flag's MIN over ranks)"""
    r, out = _run(["--gpus", "2", "--steps", "2", "--warmup", "0", "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert out["captured_allreduce_trial"][0] is True, out
    r, out = _run(["--gpus", "2", "--steps", "2", "--warmup", "0", "--dry-run", "--trial-timeout", "3"], env={"BMT_TRIAL_HANG": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    ok, why = out["captured_allreduce_trial"]
    assert ok is False and "killed" in why, out


def test_single_process_dry_run_and_world_mismatch():
    r, out = _run(["--dry-run", "--steps", "2", "--warmup", "0"])
    assert r.returncode == 0 and out["n_gpus"] == 1
    r, out = _run(["--gpus", "2", "--dry-run"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_stale_pmc_record_is_refused(tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    (tmp_path / "bmt_amd" / "csrc").mkdir(parents=True)
    (tmp_path / "include").mkdir()
    (tmp_path / "bmt_amd" / "csrc" / "k.hip").write_text("// v1\n")
    good = bench.csrc_digest()
    (prof / "r99_pmc_traffic.json").write_text(json.dumps({"csrc_digest": good, "kernels": {"gemm": {"traffic_bytes": 1.0}}}))
    rec, note = bench.pmc_record()
    assert rec is not None and good in note
    (tmp_path / "bmt_amd" / "csrc" / "k.hip").write_text("// v2\n")
    rec, note = bench.pmc_record()
    assert rec is None and "stale" in note


def test_pmc_record_is_tied_to_the_kernel_sources(tmp_path, monkeypatch):
    """bench.py takes roofline.traffic / mfma_busy from the newest profiles/*pmc_traffic.json only when the record's csrc digest is the
    tree's: a record taken with other kernels is refused with the reason in traffic_source."""
    import json
    import bench
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "csrc_digest", lambda: "aaaa")
    rec, note = bench.pmc_record()
    assert rec is None and "no PMC pass" in note
    (prof / "r99_pmc_traffic.json").write_text(json.dumps({"csrc_digest": "bbbb", "kernels": {}}))
    rec, note = bench.pmc_record()
    assert rec is None and "stale" in note and "bbbb" in note
    (prof / "r99_pmc_traffic.json").write_text(json.dumps({"csrc_digest": "aaaa", "kernels": {"gemm_w2": {"traffic_bytes": 1.0}}}))
    rec, note = bench.pmc_record()
    assert rec["kernels"]["gemm_w2"]["traffic_bytes"] == 1.0 and "aaaa" in note
    # ... and to the procedure it was taken over: a train_cap pass says nothing about train_prop's launches
    (prof / "r99_pmc_traffic.json").write_text(json.dumps({"csrc_digest": "aaaa", "procedure": "train_cap", "kernels": {"gemm_w2": {"traffic_bytes": 1.0}}}))
    assert bench.pmc_record("train_cap")[0] is not None
    rec, note = bench.pmc_record("train_prop")
    assert rec is None and "train_prop" in note


def test_kernel_classes_map_to_pmc_families():
    import bench
    fam = bench.PMC_FAMILY_OF_KERNEL
    assert fam("void (anonymous namespace)::gemm_pipe_kernel<2, true, 1, false, false>((anonymous namespace)::GemmB)") == "gemm_w2"
    assert fam("void (anonymous namespace)::gemm_pipe_kernel<2, true, 2, false, false, 1>((anonymous namespace)::GemmB)") == "conv_w2"
    assert fam("void (anonymous namespace)::gemm_pipe_kernel<2, true, 2, false, false, 0>((anonymous namespace)::GemmB)") == "gemm_w2"
    assert fam("gemm_bf16_kernel<1, 4, 1, true, true, 2, false>(GemmB)") == "conv_bf16"
    assert fam("gemm_bf16_kernel<3, 4, 1, false, false, 1, false>(GemmB)") == "conv_x3"
    assert fam("gemm_bf16_kernel<1, 4, 1, false, true, 0, false>(GemmB)") == "gemm_bf16"
    assert bench.pmc_keys_of_class("conv_planes_fp16 x (fp16 hi+lo)") == ("conv_w2",) and bench.pmc_keys_of_class("conv_planes_bf16") == ("conv_bf16",)
    assert fam("gemm_wide_kernel<true>(GemmB)") == "gemm_w2"
    assert fam("gemm_bf16_kernel<1, 4, 1, false, true, 0, false>(GemmB)") == "gemm_bf16"
    assert fam("gemm_bf16_kernel<3, 4, 1, false, false, 0, false>(GemmB)") == "gemm_x3"
    assert fam("gemm_bf16_grouped_kernel<1, 4, 1, true, true, false>(GemmB const*, XcdSeg const*, int const*)") == "gemm_dw_grouped"
    assert fam("attn_bwd_dkv32_kernel<256, false>(AttnPB)") == "attn_bwd_dkv" and fam("attn_fwd64_kernel<256, true>(AttnPB)") == "attn_fwd"
    assert bench.pmc_keys_of_class("gemm_planes_fp16 x (fp16 hi+lo)") == ("gemm_w2",)
    assert bench.pmc_keys_of_class("attn_bwd_enc_dk256_bf16") == ("attn_bwd_dq", "attn_bwd_dkv")
    assert bench.pmc_keys_of_class("gemm_planes_dw_grouped_bf16") == ("gemm_dw_grouped",)
    # the decoder's fused cross-attention launches (csrc/raw_memory.hip): one family, whatever the operand format in the class name
    assert fam("void (anonymous namespace)::raw_attn_kernel<true, true>(unsigned short const*, long)") == "raw_attn_kernel"
    for c in ("raw_attn_fused_f16", "raw_attn_fused_f16_edges", "raw_attn_fused_f16_proj", "raw_attn_fused_bf16", "raw_attn_fused_bf16_edges", "raw_attn_fused_bf16_proj"):
        assert bench.pmc_keys_of_class(c) == ("raw_attn_kernel",)


def test_rocm_smi_clock_parser_on_a_recorded_sample():
    """bench.py's engine_clock_under_load reads `rocm-smi --showclocks --showpower`: the lines of a run on the GPU box
    (profiles/r02_s_clock_under_load.txt keeps them) parse to (sclk MHz, package W); anything else parses to None"""
    import bench
    sample = ("GPU[0]\t\t: fclk clock level: 0: (1250Mhz)\nGPU[0]\t\t: mclk clock level: 0: (2000Mhz)\nGPU[0]\t\t: sclk clock level: 1: (1937Mhz)\n"
              "=================================== Power Consumption ====================================\n"
              "GPU[0]\t\t: Current Socket Graphics Package Power (W): 1356.0\nGPU[1]\t\t: sclk clock level: 1: (2100Mhz)\n")
    assert bench.parse_smi(sample) == (1937, 1356.0)
    assert bench.parse_smi(sample, gpu=1) == (2100, None)
    assert bench.parse_smi("ERROR: no AMD GPUs found\n") is None
    rec = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r02_s_clock_under_load.txt")).read()
    idle = rec.splitlines()[0].split("idle: ", 1)[1].replace(" | ", "\n")
    assert bench.parse_smi(idle) == (99, 243.0)


def test_clock_probe_never_raises_without_a_gpu():
    """no GPU here: whatever rocm-smi does (absent, error text, garbage), the probe returns None or a well-formed record and the
    steps it was given still ran"""
    import bench
    calls = []
    out = bench.clock_under_load(lambda: calls.append(1), lambda: None, seconds=0.2)
    assert calls, "the probe must drive the step"
    assert out is None or {"sclk_mhz", "package_w", "samples", "mfma_peak_at_clock_tflops", "source"} <= set(out)
