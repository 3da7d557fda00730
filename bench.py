"""bench.py -- caption tokens/s of the train_cap step (BASELINE.json metric) on N MI355X of one node.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = zero_grad -> masks -> forward -> LabelSmoothing/n_tokens -> backward -> gradient all-reduce -> Adam
(epoch_loops/captioning_epoch_loops.py:128-141) on one synthetic batch already resident in HBM.
Workload = BASELINE.json configs[1]: B=32 per GPU (weak scaling), N=2, d_model=1024, H=4, d_audio=128,
d_video=1024, d_caps=300, T_v=256, T_a=800, T_c=30, V=10000, dropout 0.1, Adam lr 5e-5, GloVe frozen.
Forward products run split-bf16 (3 MFMA passes, log-probs within 1e-3 of the fp32 reference, see
tests/test_gpu_model.py), backward products single-pass bf16; accumulation, softmax, LayerNorm, loss, Adam in fp32.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     -- the dominant kernel class, ALGORITHMIC flops / HIP-event time measured live in the timed region
  cpu_baseline -- the CPU oracle (a port of the reference, oracle/bmt_oracle.py) on the host cores, rank 0, N=1 only
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_BF16_DENSE_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: ~2.5 PF dense bf16
HBM_PEAK_GBS = 8000.0


class KernelTimer:
    """HIP-event timing of kernel classes on the stream they are launched on (torch's current stream)."""

    def __init__(self):
        self.records = {}   # class -> list of (start, end, flops, bytes)
        self.enabled = False

    def wrap(self, ops):
        timer = self
        raw_gemm, raw_afwd, raw_abwd = ops.gemm, ops.attn_fwd, ops.attn_bwd

        def gemm(A, B, C_out, M, N, K, **kw):
            if not timer.enabled:
                return raw_gemm(A, B, C_out, M, N, K, **kw)
            prec = kw.get("precision") or ops.FWD_PRECISION
            cls = f"gemm_{'kc' if kw.get('a_kc', True) else 'rc'}_{'kc' if kw.get('b_kc', True) else 'rc'}_x{prec}"
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = raw_gemm(A, B, C_out, M, N, K, **kw)
            e.record()
            timer.records.setdefault(cls, []).append((s, e, 2.0 * M * N * K, 4.0 * (M * K + N * K + M * N)))
            return r

        def attn_fwd(q, k, v, mask, H, **kw):
            if not timer.enabled:
                return raw_afwd(q, k, v, mask, H, **kw)
            B_, Sq, D = q.shape
            Sk = k.shape[1]
            prec = kw.get("precision") or ops.FWD_PRECISION
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = raw_afwd(q, k, v, mask, H, **kw)
            e.record()
            timer.records.setdefault(f"attn_fwd_dk{D // H}_x{prec}", []).append(
                (s, e, 4.0 * B_ * Sq * Sk * D, 4.0 * B_ * D * (2 * Sq + 2 * Sk)))
            return r

        def attn_bwd(q, k, v, o, do, lse, mask, H, **kw):
            if not timer.enabled:
                return raw_abwd(q, k, v, o, do, lse, mask, H, **kw)
            B_, Sq, D = q.shape
            Sk = k.shape[1]
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = raw_abwd(q, k, v, o, do, lse, mask, H, **kw)
            e.record()
            # algorithmic backward = 5 products (dV, dP, dQ, dK + S recompute) = 2.5 x forward
            timer.records.setdefault(f"attn_bwd_dk{D // H}", []).append(
                (s, e, 10.0 * B_ * Sq * Sk * D, 4.0 * B_ * D * (4 * Sq + 4 * Sk)))
            return r

        raw_gb, raw_afb, raw_abb = ops.gemm_bf16, ops.attn_fwd_bf16, ops.attn_bwd_bf16

        def gemm_bf16(A, B, C_out, **kw):
            if not timer.enabled:
                return raw_gb(A, B, C_out, **kw)
            prec = kw.get("precision") or ops.FWD_PRECISION
            akm, bkm = kw.get("a_km", False), kw.get("b_km", False)        # k-major operand: its ROWS are the reduction index
            M = A.cols if akm else A.rows
            N = B.cols if bkm else B.rows
            K = A.rows if akm else (B.rows if bkm else A.cols)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = raw_gb(A, B, C_out, **kw)
            e.record()
            nb = 2 * prec if prec == 3 else 2
            timer.records.setdefault(f"gemm_planes_x{prec}", []).append((s, e, 2.0 * M * N * K, nb * (M * K + N * K) + 4.0 * M * N))
            return r

        def attn_fwd_bf16(qh, ql, kh, kl, vh, vl, mask, H, **kw):
            if not timer.enabled:
                return raw_afb(qh, ql, kh, kl, vh, vl, mask, H, **kw)
            B_, Sq, D = qh.shape
            Sk = kh.shape[1]
            prec = kw.get("precision") or ops.FWD_PRECISION
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = raw_afb(qh, ql, kh, kl, vh, vl, mask, H, **kw)
            e.record()
            timer.records.setdefault(f"attn_fwd_planes_dk{D // H}_x{prec}", []).append(
                (s, e, 4.0 * B_ * Sq * Sk * D, (4.0 if prec == 3 else 2.0) * B_ * D * (Sq + 2 * Sk) + 4.0 * B_ * D * Sq))
            return r

        def attn_bwd_bf16(qh, kh, vh, o, do, lse, mask, H, **kw):
            if not timer.enabled:
                return raw_abb(qh, kh, vh, o, do, lse, mask, H, **kw)
            B_, Sq, D = qh.shape
            Sk = kh.shape[1]
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = raw_abb(qh, kh, vh, o, do, lse, mask, H, **kw)
            e.record()
            timer.records.setdefault(f"attn_bwd_planes_dk{D // H}", []).append(
                (s, e, 10.0 * B_ * Sq * Sk * D, B_ * D * (2.0 * (Sq + 2 * Sk) + 8.0 * Sq + 4.0 * (Sq + 2 * Sk))))
            return r

        raw_gg = ops.gemm_bf16_grouped

        def gemm_bf16_grouped(items):
            if not timer.enabled:
                return raw_gg(items)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = raw_gg(items)
            e.record()
            fl = sum(2.0 * A.cols * B.cols * A.rows for A, B, _ in items)
            by = sum(2.0 * A.rows * (A.cols + B.cols) + 4.0 * A.cols * B.cols for A, B, _ in items)
            timer.records.setdefault("gemm_planes_x1", []).append((s, e, fl, by))      # the step's weight gradients, one launch
            return r

        ops.gemm_bf16_grouped = gemm_bf16_grouped
        raw_afp, raw_abp = ops.attn_fwd_planes, ops.attn_bwd_planes

        def attn_fwd_planes(q, k, v, B_, Sq, Sk, D, mask, H, **kw):
            if not timer.enabled:
                return raw_afp(q, k, v, B_, Sq, Sk, D, mask, H, **kw)
            prec = kw.get("precision") or ops.FWD_PRECISION
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = raw_afp(q, k, v, B_, Sq, Sk, D, mask, H, **kw)
            e.record()
            side = "enc" if min(Sq, Sk) >= 128 else "dec"
            nb = 4.0 if prec == 3 else 2.0          # hi (+lo) planes in and out
            timer.records.setdefault(f"attn_fwd_{side}_dk{D // H}_x{prec}", []).append(
                (s, e, 4.0 * B_ * Sq * Sk * D, nb * B_ * D * (2 * Sq + 2 * Sk)))
            return r

        def attn_bwd_planes(q, k, v, o, do, lse, B_, Sq, Sk, D, mask, H, drop_p, biases, **kw):
            if not timer.enabled:
                return raw_abp(q, k, v, o, do, lse, B_, Sq, Sk, D, mask, H, drop_p, biases, **kw)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = raw_abp(q, k, v, o, do, lse, B_, Sq, Sk, D, mask, H, drop_p, biases, **kw)
            e.record()
            side = "enc" if min(Sq, Sk) >= 128 else "dec"
            # algorithmic backward = 5 products (S recompute, dP, dV, dK, dQ) = 2.5 x forward; bytes: q,k,v hi planes, O hi+lo,
            # dO fp32 in; dq,dk,dv plane + transposed plane out
            timer.records.setdefault(f"attn_bwd_{side}_dk{D // H}", []).append(
                (s, e, 10.0 * B_ * Sq * Sk * D, B_ * D * (2.0 * (Sq + 2 * Sk) + 8.0 * Sq + 4.0 * (Sq + 2 * Sk))))
            return r

        ops.gemm, ops.attn_fwd, ops.attn_bwd = gemm, attn_fwd, attn_bwd
        ops.gemm_bf16, ops.attn_fwd_bf16, ops.attn_bwd_bf16 = gemm_bf16, attn_fwd_bf16, attn_bwd_bf16
        ops.attn_fwd_planes, ops.attn_bwd_planes = attn_fwd_planes, attn_bwd_planes

    def summary(self):
        out = {}
        for cls, recs in self.records.items():
            ms = sum(s.elapsed_time(e) for s, e, _, _ in recs)
            fl = sum(f for _, _, f, _ in recs)
            by = sum(b for _, _, _, b in recs)
            out[cls] = {"launches": len(recs), "ms": ms, "flops": fl, "bytes": by}
        return out


def cpu_baseline_worker():
    """runs in a child process (see cpu_baseline): the reference's CPU path as restated by the oracle -- fwd + bwd + Adam
    on a bounded sample of the same workload -- prints one JSON object."""
    from bmt_amd import synthetic as syn
    from oracle import bmt_oracle as orc
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 32))          # more threads than this only add contention at these sizes
    torch.set_num_threads(cores)
    V, Tv, Ta, Tc, Bs = 10000, 256, 800, 30, 4
    cfg = syn.cfg_config1(dout_p=0.0)
    sd = orc.init_captioning_params(cfg, V, seed=0, glove=syn.make_glove(V, cfg.d_model_caps))
    p = {k: v.clone().requires_grad_(k != "emb_C.embedder.weight") for k, v in sd.items()}
    m = {k: torch.zeros_like(v) for k, v in p.items()}
    v2 = {k: torch.zeros_like(v) for k, v in p.items()}
    batch = syn.make_cap_batch(cfg, Bs, Tv, Ta, Tc, V, seed=1234)

    def step(i):
        for t in p.values():
            t.grad = None
        loss, _, ntok = orc.train_cap_loss(p, cfg, batch["feature_stacks"], batch["captions"], syn.PAD_IDX, cfg.smoothing)
        loss.backward()
        with torch.no_grad():
            for k, t in p.items():
                if t.grad is not None:
                    orc.adam_step(t, t.grad, m[k], v2[k], i, cfg.lr)
        return int(ntok)
    step(1)
    t0 = time.perf_counter()
    toks, n = 0, 0
    while n < 2 or (time.perf_counter() - t0 < 10.0 and n < 6):
        toks += step(n + 2)
        n += 1
    dt = time.perf_counter() - t0
    print(json.dumps({"value": toks / dt, "unit": "caption tokens/s", "cores": cores, "kind": "port",
                      "sample": f"{n} train steps of config[1] at B={Bs} (fwd+bwd+Adam, fp32, dropout off), "
                                f"oracle/bmt_oracle.py on torch CPU, {cores} threads, {dt / n:.2f} s/step"}))


def cpu_baseline(timeout_s=150):
    """bounded: the oracle is timed in a child process that is killed after timeout_s (the bench must never hang on it)"""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker"], capture_output=True, text=True,
                           timeout=timeout_s, env=dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES=""))
        for line in reversed(r.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"value": None, "error": (r.stderr or "no output")[-300:]}
    except subprocess.TimeoutExpired:
        return {"value": None, "error": f"cpu baseline exceeded {timeout_s}s"}


def pmc_traffic(kernel_class):
    """HBM bytes per launch of a kernel class from the newest committed PMC pass (profiles/*pmc_traffic.json, produced by
    tools/gpu_pmc_bench.sh: FETCH_SIZE / WRITE_SIZE in separate rocprofv3 passes over this same step, gfx950 correction applied).
    The counters cannot be read from inside the process that runs the step, so this is a recorded measurement, not a live one."""
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*pmc_traffic.json")))
    if not files:
        return None, "no PMC pass committed"
    try:
        with open(files[-1]) as f:
            k = json.load(f)["kernels"].get(kernel_class)
    except (OSError, ValueError, KeyError):
        return None, "unreadable PMC summary"
    if not k:
        return None, f"{os.path.basename(files[-1])} has no entry for {kernel_class}"
    return k["traffic_bytes"], (f"profiles/{os.path.basename(files[-1])}: FETCH_SIZE x2 + WRITE_SIZE, mean over {k['launches_seen']} launches of the "
                                "eagerly issued step (separate rocprofv3 --pmc passes); includes the split-K workspace traffic")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch (weak scaling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--fwd-precision", type=int, default=3, choices=[1, 3])
    ap.add_argument("--fp32-staged-gemm", action="store_true", help="A/B: use the fp32-operand GEMM (csrc/gemm.hip)")
    ap.add_argument("--no-graph", action="store_true", help="issue every kernel from Python instead of replaying one hipGraph")
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_worker:
        return cpu_baseline_worker()

    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus} (WORLD_SIZE={world})")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from bmt_amd import ops, synthetic as syn
    from bmt_amd.model.captioning_module import BiModalTransformer
    from bmt_amd.train import CaptioningTrainStep

    ops.set_precision(fwd=args.fwd_precision, bwd=1)
    ops.USE_PLANE_GEMM = not args.fp32_staged_gemm
    V, Tv, Ta, Tc, B = 10000, 256, 800, 30, args.batch
    cfg = syn.cfg_config1(dout_p=0.1)
    cfg.device = str(dev)
    torch.manual_seed(0)                                   # identical replicas on every rank
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        model = BiModalTransformer(cfg, syn.FakeTrainDataset(V, syn.make_glove(V, cfg.d_model_caps))).to(dev)
    n_params = sum(p.numel() for p in model.parameters() if p.requires_grad)
    batch = syn.make_cap_batch(cfg, B, Tv, Ta, Tc, V, seed=1234 + rank)
    fs = {k: v.to(dev) for k, v in batch["feature_stacks"].items()}      # inputs resident in HBM before timing
    caps = batch["captions"].to(dev)
    tokens_local = int((caps[:, 1:] != syn.PAD_IDX).sum())
    ops.manual_seed(1000 + rank)
    step = CaptioningTrainStep(model, cfg, syn.PAD_IDX, data_parallel=world > 1, static_grads=True)

    timer = KernelTimer()
    if not args.no_kernel_timer:
        timer.wrap(ops)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def note(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    # ---- the step is captured into two hipGraphs ({zero_grad .. backward}, {Adam}) with the eager RCCL all-reduce of the flat
    # gradient buckets between them (nothing collective is captured); eager issue of every kernel is the fallback
    mode = "eager"
    run = lambda: step(fs, caps)
    if not args.no_graph:
        try:
            step.capture(fs, caps, warmup=2)
            run = lambda: step.replay()
            mode = "hipgraph"
        except Exception as exc:      # noqa: BLE001 -- e.g. a collective that refuses capture: keep measuring, say so
            note(f"graph capture failed ({type(exc).__name__}: {exc}); falling back to eager launches")
            torch.cuda.synchronize()
    for _ in range(args.warmup):
        loss, _ = run()
    sync()
    note(f"warmup done ({args.warmup} steps, {mode})")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, _ = run()
    sync()
    dt = time.perf_counter() - t0
    final_loss = float(loss)
    note(f"timed region done: {dt / args.steps * 1e3:.2f} ms/step ({mode})")
    timer_steps = 0
    if not args.no_kernel_timer:
        # per-kernel HIP-event timing needs individual launches: the same step, eagerly issued, right after the timed region
        timer_steps = 3
        timer.enabled = True
        for _ in range(timer_steps):
            step(fs, caps)
        torch.cuda.synchronize()
        timer.enabled = False

    t = torch.tensor([dt, float(tokens_local)], dtype=torch.float64, device=dev)
    if world > 1:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt, tokens_all = float(tmax[0]), float(tsum[1])
    else:
        tokens_all = float(tokens_local)

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = tokens_all * args.steps / dt
        # algorithmic flops of the padded-dense step: 3.257 TFLOP per B=32 train step at V~10k
        flops_step = 3.257e12 * (B / 32.0) * world
        out = {
            "metric": "caption tokens/sec/node (train_cap B=32/GPU, d=1024)", "value": value, "unit": "caption tokens/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "launch_mode": mode,
            "dtype": "bf16x3 fwd / bf16 bwd MFMA, fp32 accumulate" if args.fwd_precision == 3 else "bf16",
            "data": "synthetic",
            "config": {"workload": "configs[1]: train_cap, N=2 d_model=1024 H=4 d_aud=128 d_vid=1024 d_caps=300 "
                                   "T_v=256 T_a=800 T_c=30 V=10000, dropout 0.1, Adam, GloVe frozen",
                       "global_batch": B * world, "per_gpu_batch": B, "parallelism": f"dp{world}",
                       "trainable_params": n_params, "tokens_per_step": tokens_all, "final_loss": final_loss},
            "algorithmic_tflops": flops_step / (ms_per_step * 1e-3) / 1e12,
            "mfma_peak_frac": flops_step / (ms_per_step * 1e-3) / 1e12 / (MFMA_BF16_DENSE_PEAK_TFLOPS * world),
        }
        if not args.no_kernel_timer:
            summ = timer.summary()
            tot = sum(v["ms"] for v in summ.values()) or 1.0
            dom = max(summ, key=lambda k: summ[k]["ms"])
            d = summ[dom]
            ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
            traffic, traffic_note = pmc_traffic(dom)
            out["roofline"] = {"kernel": dom, "bound": "mfma", "achieved": ach, "peak": MFMA_BF16_DENSE_PEAK_TFLOPS,
                               "unit": "TFLOP/s", "frac": ach / MFMA_BF16_DENSE_PEAK_TFLOPS,
                               "frac_issued": ach * (3 if dom.endswith("x3") else 1) / MFMA_BF16_DENSE_PEAK_TFLOPS, "traffic": traffic,
                               "traffic_unit": "HBM bytes per launch", "traffic_source": traffic_note,
                               "algorithmic_bytes_per_launch": d["bytes"] / d["launches"] if "bytes" in d else None,
                               "launches": d["launches"], "avg_launch_us": d["ms"] * 1e3 / d["launches"],
                               "share_of_timed_kernels": d["ms"] / tot,
                               "mfma_passes": 3 if dom.endswith("x3") else 1, "gemm_path": "planes" if ops.USE_PLANE_GEMM else "fp32-staged",
                               "timing": f"HIP events around every launch of the class (a split-K GEMM launch = main kernel + its epilogue kernel), {timer_steps} eagerly issued steps right after the timed region"}
            # the north-star quantity: bi-modal ENCODER attention against the MFMA roofline.  "issued" counts what the matrix
            # pipe executes (forward: 3 split-bf16 passes; backward: 7 products as scheduled -- S and dP are computed by both
            # backward kernels), "algorithmic" the 2 / 5 products of the math.
            enc = {k: v for k, v in summ.items() if k.startswith(("attn_fwd_enc", "attn_bwd_enc"))}
            if enc:
                ms = sum(v["ms"] for v in enc.values())
                alg = sum(v["flops"] for v in enc.values())
                issued = sum(v["flops"] * (3.0 if k.startswith("attn_fwd") and k.endswith("x3") else (1.4 if k.startswith("attn_bwd") else 1.0))
                             for k, v in enc.items())
                out["attention_roofline"] = {
                    "scope": "encoder self- and cross-attention cores, forward + backward, B=32 H=4 d_k=256 T_v=256 T_a=800",
                    "bound": "mfma", "peak": MFMA_BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "algorithmic": alg / (ms * 1e-3) / 1e12, "issued": issued / (ms * 1e-3) / 1e12,
                    "frac_algorithmic": alg / (ms * 1e-3) / 1e12 / MFMA_BF16_DENSE_PEAK_TFLOPS,
                    "frac_issued": issued / (ms * 1e-3) / 1e12 / MFMA_BF16_DENSE_PEAK_TFLOPS,
                    "ms_per_step": ms / timer_steps, "share_of_timed_kernels": ms / tot}
            out["kernel_classes"] = {k: {"ms_per_step": v["ms"] / timer_steps, "tflops": v["flops"] / (v["ms"] * 1e-3) / 1e12,
                                         "gbs": v["bytes"] / (v["ms"] * 1e-3) / 1e9, "launches_per_step": v["launches"] / timer_steps}
                                     for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])}
        if world == 1 and not args.no_cpu_baseline:
            note("timing the CPU oracle (bounded sample, child process)")
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
