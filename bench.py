"""bench.py -- caption tokens/s of the train_cap step (BASELINE.json metric) on N MI355X of one node.

    python bench.py --gpus N --steps K --warmup W          (N > 1: re-executes itself under torch.distributed.run, one rank per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W             (the driver's form: RANK / LOCAL_RANK / WORLD_SIZE from the environment)
    python bench.py --procedure train_prop                 (configs[3]: the proposal generator step, its own roofline line)

A step = zero_grad -> masks -> forward -> LabelSmoothing/n_tokens -> backward -> gradient all-reduce -> Adam
(epoch_loops/captioning_epoch_loops.py:128-141) on one synthetic batch already resident in HBM.
Workload = BASELINE.json configs[1]: B=32 per GPU (weak scaling), N=2, d_model=1024, H=4, d_audio=128,
d_video=1024, d_caps=300, T_v=256, T_a=800, T_c=30, V=10000, dropout 0.1, Adam lr 5e-5, GloVe frozen.
Forward products follow the per-site operand policy of bmt_amd/ops.py (DESIGN.md "precision": fp16 / split-fp16 / split-bf16
MFMA operands chosen so that the log-probs stay within 1e-3 of the fp32 reference, see tests/test_gpu_model.py), backward
products run single-pass bf16; accumulation, softmax, LayerNorm, loss, Adam in fp32.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     -- the dominant kernel class, ALGORITHMIC flops / HIP-event time measured live in the timed region
  cpu_baseline -- the CPU oracle (a port of the reference, oracle/bmt_oracle.py) on the host cores, rank 0, N=1 only
"""
import argparse
import json
import math
import os
import sys
import time

# kernel arguments in device memory (the HIP runtime's default on this ROCm; stated here so that an environment that switches it off does
# not silently cost the ~400 small launches of a step 0.4 ms: measured 9.11 vs 8.73 ms/step with it forced off, same box)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_BF16_DENSE_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: ~2.5 PF dense bf16
HBM_PEAK_GBS = 8000.0


class KernelTimer:
    """HIP-event timing of kernel classes on the stream they are launched on (torch's current stream).

    One event pair around every launch of a class, over several eagerly issued steps.  A class's time per step is the sum over its
    launches of the MEDIAN over the steps of that launch's interval (the launch sequence of a step is deterministic: launch i of
    step k is the same kernel on the same shapes as launch i of every other step), so a host stall that lands between an event and
    its kernel in one step (an allocator miss, a page fault, a descheduled Python thread) does not reach the number -- the sum of raw
    intervals did (round 3's driver run: 13.5 ms of "attention backward" inside an 8.7 ms step)."""

    def __init__(self):
        self.steps = []     # per eager step: list of (class, start event, end event, flops, bytes, executed flops, executed bytes)
        self.passes = {}    # class -> MFMA passes per algorithmic product
        self.enabled = False
        # what of the padded batch exists (round 6): rows = {capacity rows of a packed stream: valid rows of the timer's batch},
        # lens = {padded sequence length: tensor of per-sample valid lengths}.  With them every class also carries the FLOPs and bytes of the
        # rows the kernels actually process ("executed"); the padded-dense figures of SURVEY.md 8d stay beside them.
        self.valid_rows = {}
        self.valid_lens = {}

    def set_valid(self, rows, lens):
        self.valid_rows, self.valid_lens = dict(rows), dict(lens)

    def xattn(self, B, Sq, Sk, qpacked, kpacked):
        """(sum_b Lq_b Lk_b, sum_b Lq_b, sum_b Lk_b) of an attention launch: what exists of its (B, Sq, Sk) problem.  A padded side counts in full
        (the kernels skip masked key TILES and dead query tiles there too; the count stays the padded one)."""
        lq = self.valid_lens.get(Sq) if qpacked else None
        lk = self.valid_lens.get(Sk) if kpacked else None
        lq = [float(x) for x in lq] if lq is not None and len(lq) == B else [float(Sq)] * B
        lk = [float(x) for x in lk] if lk is not None and len(lk) == B else [float(Sk)] * B
        return sum(a * b for a, b in zip(lq, lk)), sum(lq), sum(lk)

    def xrows(self, n, packed):
        """rows of a packed stream that exist (the capacity n where the layout is padded or unknown)"""
        return self.valid_rows.get(n, n) if packed else n

    def begin_step(self):
        self.steps.append([])

    def _timed(self, cls, passes, flops, nbytes, fn, xflops=None, xbytes=None):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        r = fn()
        e.record()
        if not self.steps:
            self.steps.append([])
        self.steps[-1].append((cls, s, e, flops, nbytes, flops if xflops is None else xflops, nbytes if xbytes is None else xbytes))
        self.passes[cls] = passes
        return r

    def reset(self):
        self.steps = []

    def wrap(self, ops):
        timer = self
        raw_gb, raw_gg = ops.gemm_bf16, ops.gemm_bf16_grouped
        raw_afp, raw_abp = ops.attn_fwd_planes, ops.attn_bwd_planes

        def gemm_bf16(A, B, C_out, **kw):
            if not timer.enabled:
                return raw_gb(A, B, C_out, **kw)
            prec = kw["precision"]
            akm, bkm, conv = kw.get("a_km", False), kw.get("b_km", False), kw.get("conv")
            if conv is not None and conv["mode"] == 1:       # implicit Conv1d forward / dX: reduction over (tap, channel)
                M, N, K = conv["M"], B.rows, B.any.shape[1]
            elif conv is not None:                            # implicit Conv1d dW
                M, N, K = A.cols, conv["N"], A.rows
            else:                                             # k-major operand: its ROWS are the reduction index
                M = A.cols if akm else A.rows
                N = B.cols if bkm else B.rows
                K = A.rows if akm else (B.rows if bkm else A.cols)
                blk = kw.get("a_blk")                         # a block product (ops.gemm_bf16): every output block reduces over ITS a_blk[1] indices
                if blk is not None:
                    if bkm:
                        N = (K // blk[1]) * blk[0]
                    K = blk[1]
            nb = ops.prec_operand_bytes(prec)
            cls = ("conv_planes_" if conv is not None else "gemm_planes_") + ops.prec_name(prec)
            op = kw.get("out_planes")          # output bytes per element: 4 for the fp32 tensor, 2 per 16-bit plane actually written
            ob = (4.0 if C_out is not None else 0.0) + (2.0 * sum(t is not None for t in (op.hi, op.lo, op.fh)) if op is not None else 0.0)
            # executed: a packed activation operand bounds the product's rows (a weight gradient: its reduction) by the rows that exist
            pk = conv is None and (A.pack is not None or (akm and bkm and B.pack is not None))
            xM = M if akm else timer.xrows(M, pk)
            xK = timer.xrows(K, pk) if akm else K
            return timer._timed(cls, ops.prec_passes(prec), 2.0 * M * N * K, nb[0] * M * K + nb[1] * N * K + ob * M * N,
                                lambda: raw_gb(A, B, C_out, **kw),
                                xflops=2.0 * xM * N * xK, xbytes=nb[0] * xM * xK + nb[1] * N * xK + ob * xM * N)

        def gemm_bf16_grouped(items, **kw):
            if not timer.enabled:
                return raw_gg(items, **kw)
            fl = sum(2.0 * it[0].cols * it[1].cols * it[0].rows for it in items)
            by = sum(2.0 * it[0].rows * (it[0].cols + it[1].cols) + 4.0 * it[0].cols * it[1].cols for it in items)
            xr = [timer.xrows(it[0].rows, it[0].pack is not None or it[1].pack is not None) for it in items]     # the reduction over the rows that exist
            xfl = sum(2.0 * it[0].cols * it[1].cols * r for it, r in zip(items, xr))
            xby = sum(2.0 * r * (it[0].cols + it[1].cols) + 4.0 * it[0].cols * it[1].cols for it, r in zip(items, xr))
            if len(items[0]) > 3:                  # the gradient of an encoder memory (ops.RawMemoryFn): one product per sample, packed output rows
                return timer._timed("gemm_planes_memory_grad_grouped_bf16", 1, fl, by, lambda: raw_gg(items, **kw))
            return timer._timed("gemm_planes_dw_grouped_bf16", 1, fl, by, lambda: raw_gg(items, **kw), xflops=xfl, xbytes=xby)     # the step's weight gradients, one launch

        def _rank_dims(D, H, kw):
            """an attention launch in the rank form (ops.RankSelfAttnFn / RankCrossAttnFn: kv_shared, the softmax scale of the reference's head size):
            (model width of the reference's formulation -- what the algorithmic FLOPs / bytes count --, width of the key / value plane read, class tag)"""
            if not kw.get("kv_shared"):
                return D, D, ""
            dk_ref = int(round(1.0 / float(kw["scale"]) ** 2)) if kw.get("scale") else D // H
            return H * dk_ref, D // H, f"_rank_of_dk{dk_ref}"

        def attn_fwd_planes(q, k, v, B_, Sq, Sk, D, mask, H, **kw):
            if not timer.enabled:
                return raw_afp(q, k, v, B_, Sq, Sk, D, mask, H, **kw)
            prec = kw.get("precision", ops.PREC_BF16X3)
            side = "enc" if min(Sq, Sk) >= 128 else "dec"
            nb = sum(ops.prec_operand_bytes(prec)) / 2.0          # operand plane bytes per element in, the same again out
            xqk, xq, xk = timer.xattn(B_, Sq, Sk, q.pack is not None, k.pack is not None)
            Dref, Dkv, tag = _rank_dims(D, H, kw)
            return timer._timed(f"attn_fwd_{side}_dk{D // H}{tag}_{ops.prec_name(prec)}", ops.prec_passes(prec),
                                4.0 * B_ * Sq * Sk * Dref, nb * B_ * Dref * (Sq + 2 * Sk) + 4.0 * B_ * Dref * Sq,
                                lambda: raw_afp(q, k, v, B_, Sq, Sk, D, mask, H, **kw),
                                xflops=4.0 * xqk * D, xbytes=nb * (D * xq + 2 * Dkv * xk) + 4.0 * D * xq)

        def attn_bwd_planes(q, k, v, o, do, lse, B_, Sq, Sk, D, mask, H, drop_p, biases, **kw):
            if not timer.enabled:
                return raw_abp(q, k, v, o, do, lse, B_, Sq, Sk, D, mask, H, drop_p, biases, **kw)
            side = "enc" if min(Sq, Sk) >= 128 else "dec"
            # algorithmic backward = 5 products (S recompute, dP, dV, dK, dQ) = 2.5 x forward; bytes: q,k,v bf16 planes, O planes,
            # dO plane in; dq,dk,dv planes out
            split = ops.ATTN_BWD_SPLIT and q.hi is None and Sq >= 64 and D // H >= 128       # (ops._attn_split_ws / bmt_attn_bwd_split_ws)
            form = ("f16+bf16_recompute" if ops.ATTN_BWD_RECOMPUTE and Sq <= 2048 else "f16+bf16_split") if split else "bf16"
            xqk, xq, xk = timer.xattn(B_, Sq, Sk, q.pack is not None, k.pack is not None)
            Dref, Dkv, tag = _rank_dims(D, H, kw)
            return timer._timed(f"attn_bwd_{side}_dk{D // H}{tag}_" + form, 1, 10.0 * B_ * Sq * Sk * Dref,
                                B_ * Dref * (2.0 * (Sq + 2 * Sk) + 6.0 * Sq + 2.0 * (Sq + 2 * Sk)),
                                lambda: raw_abp(q, k, v, o, do, lse, B_, Sq, Sk, D, mask, H, drop_p, biases, **kw),
                                xflops=10.0 * xqk * D, xbytes=2.0 * (D * xq + 2 * Dkv * xk) + 6.0 * D * xq + 2.0 * D * (xq + 2 * xk))

        raw_gbt = ops.gemm_batched

        def gemm_batched(prec, M, N, Kpad, nb_o, nb_i, *a, **kw):
            if not timer.enabled:
                return raw_gbt(prec, M, N, Kpad, nb_o, nb_i, *a, **kw)
            nb, n = ops.prec_operand_bytes(prec), nb_o * nb_i
            ob = (4.0 if kw.get("C_") else 0.0) + 2.0 * sum(kw.get(k) is not None for k in ("p1", "p2"))
            return timer._timed("gemm_small_batched_" + ops.prec_name(prec), ops.prec_passes(prec), 2.0 * M * N * Kpad * n,
                                n * (nb[0] * M * Kpad + nb[1] * N * Kpad + ob * M * N), lambda: raw_gbt(prec, M, N, Kpad, nb_o, nb_i, *a, **kw))

        raw_ral = ops.raw_attn_launch

        def raw_attn_launch(bwd, B_, H, Tq, S, dm, fn, edges_dk=0, proj_k=0):
            if not timer.enabled:
                return raw_ral(bwd, B_, H, Tq, S, dm, fn, edges_dk=edges_dk, proj_k=proj_k)
            # two products of (H Tq) x S x dm per sample; bytes: the A rows and the result rows (16-bit), the memory both ways, P / dS (16-bit, twice)
            fl = 2 * 2.0 * B_ * H * Tq * S * dm
            by = B_ * (2.0 * 2 * H * Tq * dm + 2.0 * 2 * S * dm + 2.0 * 2 * H * Tq * S)
            if edges_dk:      # + the block products either side (H Tq x dm x d_k each: two behind / in front of the backward, one -- split-bf16 -- in front
                n_e = 2 if bwd else 1      # of the forward), their d_k-wide rows and the weight blocks
                fl += n_e * 2.0 * B_ * H * Tq * dm * edges_dk
                by += B_ * 2.0 * n_e * (1 if bwd else 2) * H * Tq * edges_dk + 2.0 * n_e * (1 if bwd else 2) * H * edges_dk * dm
            if proj_k:        # + the query projection / the out-projection's dX (H Tq x d_k x proj_k): the stream's rows, the weight's planes, the d_k-wide result
                fl += 2.0 * B_ * H * Tq * edges_dk * proj_k
                by += B_ * 2.0 * 2 * Tq * proj_k + 2.0 * 2 * H * edges_dk * proj_k + B_ * 2.0 * H * Tq * edges_dk
            return timer._timed("raw_attn_fused_" + ("bf16" if bwd else "f16") + ("_proj" if proj_k else "_edges" if edges_dk else ""), 1, fl, by, fn)

        ops.raw_attn_launch = raw_attn_launch
        ops.gemm_batched = gemm_batched
        ops.gemm_bf16, ops.gemm_bf16_grouped = gemm_bf16, gemm_bf16_grouped
        ops.attn_fwd_planes, ops.attn_bwd_planes = attn_fwd_planes, attn_bwd_planes

    def summary(self):
        """class -> {launches (per step), ms (per step), flops, bytes (per step), spread}; ``ms`` = sum over the class's launches of
        the median over the steps of the launch's interval (see the class comment).  Steps whose launch sequence differs from the first
        step's (never seen: the step is static) are left out and counted in ``aligned_steps``."""
        return summarize_intervals([[(c, s.elapsed_time(e), f, b, xf, xb) for c, s, e, f, b, xf, xb in st] for st in self.steps])


def summarize_intervals(steps):
    """steps: per eager step a list of (class, interval ms, flops, bytes) in launch order -> (per-class summary, steps used).  Pure
    host arithmetic (tests/test_bench_timer.py)."""
    import statistics
    steps = [st for st in steps if st]
    if not steps:
        return {}, 0
    seq = [c for c, *_ in steps[0]]
    aligned = [st for st in steps if [c for c, *_ in st] == seq]
    out = {}
    for i, item in enumerate(aligned[0]):
        cls, _, fl, by = item[:4]
        xfl, xby = (item[4], item[5]) if len(item) > 5 else (fl, by)       # executed (valid rows); the padded-dense figures where not given
        xs = [st[i][1] for st in aligned]
        d = out.setdefault(cls, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0, "ms_sum_of_means": 0.0, "ms_max_step": 0.0,
                                 "xflops": 0.0, "xbytes": 0.0})
        d["launches"] += 1
        d["ms"] += statistics.median(xs)
        d["ms_sum_of_means"] += sum(xs) / len(xs)
        d["flops"] += fl
        d["bytes"] += by
        d["xflops"] += xfl
        d["xbytes"] += xby
    for cls, d in out.items():          # the worst single step of the class: what a sum of raw intervals would have been pulled towards
        d["ms_max_step"] = max(sum(x[1] for x in st if x[0] == cls) for st in aligned)
    return out, len(aligned)


def cap_step_flops(La, Lv, Ta=800, Tv=256, layers=2):
    """algorithmic FLOPs of one configs[1] train_cap step (3 x forward; SURVEY.md 8d's per-sample, per-layer MFLOP figures) over the rows
    that EXIST: a sample with La of Ta audio and Lv of Tv video positions costs the row-wise encoder products in proportion to its rows,
    the attention cores in proportion to Lq * Lk, the decoder's products against an encoder memory in proportion to the memory's length;
    caption rows are not packed and count in full.  La = Ta, Lv = Tv for every sample gives the padded-dense 3.257 TFLOP per 32 samples."""
    tot = 0.0
    for la, lv in zip(La, Lv):
        a, v = float(la) / Ta, float(lv) / Tv
        enc = (838.9 * a + 2147.5 * v                                   # self-attention projections (audio, video)
               + (2 * 209.7 * a + 1073.7 * v) + (2 * 536.9 * v + 419.4 * a)      # cross-attention q / out of the querying stream, k / v of the other
               + 2621.4 * a * a + 268.4 * v * v + 2 * 838.9 * a * v          # the four attention cores
               + 209.7 * a + 4295.0 * v)                                # FFNs
        dec = (73.7 + 3.7 + (456.3 - 419.4) + 419.4 * a + 98.3 * a + (1110.6 - 1073.7) + 1073.7 * v + 31.5 * v + 10.8 + 43.2)
        tot += layers * (enc + dec) + 183.1
    return 3.0 * tot * 1e6


def roofline_gates(classes_ms, eager_ms, ms_per_step, eager_slack=1.6):
    """the sanity gates a kernel-timer pass must clear before its numbers go into the line: every violated gate as a
    sentence, [] when the pass is consistent.  classes_ms: class -> ms per step; eager_ms: the eagerly issued one-stream step (median of
    the per-step event intervals); ms_per_step: the timed region's (graph, two streams) step."""
    bad = []
    tot = sum(classes_ms.values())
    if not (eager_ms and eager_ms > 0):
        return ["no eager step time"]
    if tot > eager_ms * 1.02:
        bad.append(f"timed classes sum to {tot:.2f} ms/step, more than the eager step they were timed in ({eager_ms:.2f} ms)")
    if eager_ms > eager_slack * ms_per_step:
        bad.append(f"the eager one-stream step took {eager_ms:.2f} ms, more than {eager_slack} x the timed region's {ms_per_step:.2f} ms/step")
    for k, v in classes_ms.items():
        if v > ms_per_step:
            bad.append(f"class {k} alone takes {v:.2f} ms/step, more than the whole step ({ms_per_step:.2f} ms)")
    return bad


def cpu_baseline_worker():
    """runs in a child process (see cpu_baseline): the reference's CPU path as restated by the oracle -- the identical config[1]
    step (B=32, fwd + bwd + Adam, fp32, dropout 0.1) -- 3 warm-up steps, then timed steps; prints one JSON object after EVERY
    timed step (median so far), so a parent that has to kill the child on its time bound still has a number."""
    from bmt_amd import synthetic as syn
    from oracle import bmt_oracle as orc
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 64))
    if "--threads" in sys.argv:
        cores = max(1, min(avail, int(sys.argv[sys.argv.index("--threads") + 1])))
    torch.set_num_threads(cores)
    V, Tv, Ta, Tc, Bs = 10000, 256, 800, 30, 32
    cfg = syn.cfg_config1(dout_p=0.1)
    orc.set_dropout(cfg.dout_p)
    torch.manual_seed(0)
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
    warm, timed = 3, 5
    t_w = time.perf_counter()
    for i in range(warm):
        step(i + 1)
        if time.perf_counter() - t_w > 60.0 and i >= 0:      # a slow host: spend the bound on timed steps
            warm = i + 1
            break
    times, toks = [], 0
    for n in range(timed):
        t0 = time.perf_counter()
        toks = step(warm + n + 1)
        times.append(time.perf_counter() - t0)
        med = sorted(times)[len(times) // 2]
        print(json.dumps({"value": toks / med, "unit": "caption tokens/s", "cores": cores, "kind": "port",
                          "sample": f"config[1] step at B={Bs} (fwd+bwd+Adam, fp32, dropout {cfg.dout_p}), oracle/bmt_oracle.py on torch "
                                    f"{torch.__version__} CPU, {cores} threads: {warm} warm-up + {len(times)} timed steps, median {med:.2f} s/step, "
                                    f"{toks} tokens/step"}), flush=True)


def physical_cores():
    """distinct (package, core) pairs of the CPUs this process may run on (hyper-threads of a core count once)"""
    try:
        allowed = os.sched_getaffinity(0)
        seen = set()
        for c in allowed:
            base = f"/sys/devices/system/cpu/cpu{c}/topology/"
            with open(base + "physical_package_id") as f1, open(base + "core_id") as f2:
                seen.add((f1.read().strip(), f2.read().strip()))
        return max(1, len(seen))
    except (OSError, AttributeError):
        return max(1, (os.cpu_count() or 2) // 2)


def cpu_baseline(timeout_s=240):
    """the oracle on the host cores at TWO thread counts -- one thread per physical core and (as in rounds 1-2) min(logical cores, 64) --
    and the better of the two is the baseline (the survey timed the actual reference at 8 threads on 8 cores; 64 threads on a box with
    fewer physical cores oversubscribes MKL).  Each run is a bounded child process."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    counts = sorted({max(1, min(avail, 64)), max(1, min(avail, physical_cores(), 64))})
    runs = [_cpu_baseline_run(timeout_s // len(counts), n) for n in counts]
    good = [r for r in runs if r.get("value")]
    if not good:
        return runs[-1]
    best = max(good, key=lambda r: r["value"])
    if len(runs) > 1:
        best["sample"] += "; thread counts tried: " + ", ".join(f"{r.get('cores')} -> {r['value']:.1f} tokens/s" if r.get("value") else f"{r.get('cores')} -> no result" for r in runs)
    # what the number is worth: the PORT is slower than the thing it stands in for -- the reference itself, imported unmodified, ran a step of
    # this configuration in 12.07 s on 8 cores of the build container (survey time).  On THIS batch's token count:
    toks = None
    import re
    m = re.search(r"(\d+) tokens/step", best.get("sample", ""))
    if m:
        toks = int(m.group(1))
        best["reference_itself"] = {"s_per_step": 12.07, "cores": 8, "tokens_per_step": toks, "tokens_per_s_on_this_batch": toks / 12.07,
                                    "where": "build container, survey time (BASELINE.md section 2: 12.07 s per B=32 step); never runs on the GPU box"}
    best["sample"] += "; a stated baseline, not a target"
    return best


def _cpu_baseline_run(timeout_s, threads):
    """bounded: the oracle is timed in a child process that is killed after timeout_s (the bench must never hang on it); the
    child reports after every timed step, the last report wins"""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", "--threads", str(threads)]
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out, err, note = "", "", None
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env)
        out, err = r.stdout, r.stderr
    except subprocess.TimeoutExpired as exc:
        out = exc.stdout.decode() if isinstance(exc.stdout, bytes) else (exc.stdout or "")
        note = f"stopped at the {timeout_s} s bound"
    for line in reversed(out.strip().splitlines()):
        if line.startswith("{"):
            res = json.loads(line)
            if note:
                res["sample"] += f" ({note})"
            return res
    return {"value": None, "cores": threads, "error": (note or err or "no output")[-300:]}


def csrc_digest():
    """sha256 over the HIP sources + the C ABI header: ties a recorded PMC pass to the kernels it measured"""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "bmt_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "bmt_amd", "csrc", "*.h"))
                    + glob.glob(os.path.join(ROOT, "include", "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


NOMINAL_SCLK_MHZ = 2400.0     # the engine clock MFMA_BF16_DENSE_PEAK_TFLOPS is quoted at (MI355X_MICROARCH.md)


def parse_smi(text: str, gpu: int = 0):
    """(sclk MHz, package W or None) of GPU[gpu] from `rocm-smi --showclocks --showpower` output, or None"""
    import re
    sclk = power = None
    for ln in text.splitlines():
        if not ln.startswith(f"GPU[{gpu}]"):
            continue
        m = re.search(r"sclk clock level:.*\((\d+)Mhz\)", ln)
        if m:
            sclk = int(m.group(1))
        m = re.search(r"Power \(W\):\s*([\d.]+)", ln)
        if m:
            power = float(m.group(1))
    return None if sclk is None else (sclk, power)


def clock_under_load(run, sync, seconds=2.0):
    """engine clock and package power while the step runs back to back, AFTER the timed region: the matrix loops hold the package at
    its power limit and pay with clock, so the 2.4 GHz `peak` is not what the kernels can see.  rocm-smi polled
    from a side thread; every failure mode (no rocm-smi, no permission, unparsable output) yields None -- the line never depends on it."""
    import statistics
    import subprocess
    import threading
    samples, stop = [], threading.Event()

    def poll():
        while not stop.is_set():
            try:
                out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
                smp = parse_smi(out)
            except Exception:      # noqa: BLE001
                return
            if smp is None:
                return
            samples.append(smp)

    try:
        th = threading.Thread(target=poll, daemon=True)
        th.start()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            for _ in range(10):
                run()
            sync()
        stop.set()
        th.join(timeout=15)
        use = samples[1:] if len(samples) >= 3 else samples      # the first poll can catch the ramp from idle
        if not use:
            return None
        sclk = statistics.median(x[0] for x in use)
        pw = [x[1] for x in use if x[1] is not None]
        return {"sclk_mhz": sclk, "package_w": statistics.median(pw) if pw else None, "samples": len(use),
                "mfma_peak_at_clock_tflops": MFMA_BF16_DENSE_PEAK_TFLOPS * sclk / NOMINAL_SCLK_MHZ,
                "source": f"rocm-smi --showclocks --showpower polled while the step ran back to back for {seconds:.0f} s after the timed "
                          f"region; `peak` elsewhere in this line stays the nominal {MFMA_BF16_DENSE_PEAK_TFLOPS:.0f} TFLOP/s at "
                          f"{NOMINAL_SCLK_MHZ:.0f} MHz"}
    except Exception:      # noqa: BLE001
        return None


def PMC_FAMILY_OF_KERNEL(name: str) -> str:
    """kernel name in a rocprofv3 trace -> the family key of profiles/*pmc_traffic.json (tools/pmc_traffic.py)"""
    import re
    n = re.sub(r"\(anonymous namespace\)::", "", name)
    n = re.sub(r"^void ", "", n)
    if n.startswith("gemm_bf16_grouped_kernel"):
        return "gemm_dw_grouped"
    if n.startswith("gemm_memory_grad_kernel"):       # the same body over the per-sample products of an encoder memory's gradient (ops.RawMemoryFn)
        return "gemm_memory_grad"
    m = re.match(r"gemm_small_kernel<(\d), (true|false)", n)      # 32 x 32 tiles: the decoder's small products, plain and batched
    if m:
        return "gemm_small_x3" if m.group(1) == "3" else ("gemm_small_f16" if m.group(2) == "true" else "gemm_small_bf16")
    # implicit Conv1d launches (CONV, the sixth template argument, is 1 or 2): families of their own -- the proposal heads' k-tap products
    m = re.match(r"gemm_(?:bf16_kernel<(\d), \d, \d, (?:true|false), (?:true|false), ([12])|pipe_kernel<(\d), (?:true|false), \d, (?:true|false), (?:true|false), ([12]))", n)
    if m:
        npass = m.group(1) or m.group(3)
        f16 = ("true>" in n.split("(")[0][-8:]) if n.startswith("gemm_bf16_kernel") else bool(re.match(r"gemm_pipe_kernel<\d, true", n))
        return {"1": "conv_f16" if f16 else "conv_bf16", "2": "conv_w2", "3": "conv_x3"}[npass]
    if re.match(r"gemm_bf16_kernel<1, \d, \d, false, false, \d, true", n) or re.match(r"gemm_pipe_kernel<1, true", n):
        return "gemm_f16"
    m = re.match(r"gemm_(?:bf16|pipe)_kernel<(\d)", n)
    if m:
        return {"1": "gemm_bf16", "2": "gemm_w2", "3": "gemm_x3"}[m.group(1)]
    if n.startswith("gemm_wide_kernel<true"):
        return "gemm_w2"            # the step's fp16 products are all two-plane (ops.POLICIES); a one-plane fp16 launch would land here too
    if n.startswith("gemm_wide_kernel<false"):
        return "gemm_bf16"
    m = re.match(r"gemm_k128_kernel<(true|false), (true|false)", n)          # <F16, TWO planes, ...>: the reduction-of-128 products
    if m:
        return "gemm_w2" if m.group(2) == "true" else ("gemm_f16" if m.group(1) == "true" else "gemm_bf16")
    if n.startswith("attn_fwd"):
        return "attn_fwd"
    if n.startswith("attn_bwd_pair"):        # the two-kernel form's dQ and dK / dV workgroups in one launch (the decoder's attentions)
        return "attn_bwd_pair"
    if n.startswith("attn_bwd_dq"):
        return "attn_bwd_dq"
    if n.startswith("attn_bwd_dkv"):
        return "attn_bwd_dkv"
    return n.split("(")[0].split("<")[0]


# kernel class of the KernelTimer -> family key(s) of the PMC record (a backward attention launch is the dQ and the dK / dV kernel)
def pmc_keys_of_class(cls: str):
    if cls.startswith("raw_attn_fused"):       # the decoder's cross-attention middle against a raw memory, one launch (csrc/raw_memory.hip)
        return ("raw_attn_kernel",)
    if cls.startswith("attn_fwd"):
        return ("attn_fwd",)
    if cls.startswith("attn_bwd"):
        return ("attn_bwd_dq", "attn_bwd_dkv")
    if cls.startswith("conv_"):
        if cls.endswith("bf16x3"):
            return ("conv_x3",)
        if "(fp16 hi+lo)" in cls:
            return ("conv_w2",)
        return ("conv_f16",) if cls.endswith("_fp16") else ("conv_bf16",)
    if "dw_grouped" in cls:
        return ("gemm_dw_grouped",)
    if "memory_grad" in cls:
        return ("gemm_memory_grad",)
    if cls.startswith("gemm_small_batched"):
        return ("gemm_small_x3",) if cls.endswith("bf16x3") else (("gemm_small_f16",) if cls.endswith("_fp16") else ("gemm_small_bf16",))
    if cls.endswith("bf16x3"):
        return ("gemm_x3", "gemm_small_x3")      # (the class is by operand format: the decoder's products on 32 x 32 tiles + the generator's on 128-row tiles)
    if "(fp16 hi+lo)" in cls:
        return ("gemm_w2",)
    if cls.endswith("_fp16"):
        return ("gemm_f16",)
    if cls.endswith("_bf16"):
        return ("gemm_bf16", "gemm_small_bf16")
    return (cls,)


def pmc_record(procedure=None):
    """the newest committed PMC pass (profiles/*pmc_traffic.json, written by tools/gpu_pmc_bench.sh: FETCH_SIZE / WRITE_SIZE / SQ
    counters in separate rocprofv3 --pmc passes over this same step, gfx950 FETCH correction applied).  The counters cannot be
    read from inside the process that runs the step, so this is a recorded measurement; it is REFUSED when the kernel sources
    have changed since (the record carries the digest of bmt_amd/csrc it was taken with)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")))
    if not files:
        return None, "no PMC pass committed"
    # a record is a pass over ONE procedure's step (the other procedure's launches -- other shapes, the Conv1d kernels -- are not in it): the
    # newest one taken over this procedure
    # ... taken with THIS tree's kernel sources (file names are tags, not dates: "r05_end" sorts before "r05_z")
    rec = name = None
    have = csrc_digest()
    newest = None
    for path in reversed(files):
        try:
            with open(path) as f:
                r = json.load(f)
        except (OSError, ValueError):
            if path == files[-1] and procedure is None:
                return None, f"profiles/{os.path.basename(path)}: unreadable"
            continue
        if procedure is None or r.get("procedure") in (None, procedure):
            if newest is None:
                newest = (r, os.path.basename(path))
            if r.get("csrc_digest") == have:
                rec, name = r, os.path.basename(path)
                break
    if rec is None and newest is not None:
        rec, name = newest
    if rec is None:
        return None, f"no PMC pass over the {procedure} step committed (tools/gpu_pmc_bench.sh <tag> {procedure})"
    want = rec.get("csrc_digest")
    if want != have:
        return None, f"profiles/{name} was taken with kernel sources {want}, the tree is {have}: stale, refused"
    return rec, f"profiles/{name} (csrc {have}): FETCH_SIZE x2 + WRITE_SIZE per launch, separate rocprofv3 --pmc passes over the eagerly issued step"


def graph_overlap_trial(args, world, rank, local_rank, dev, timeout_s=None):
    """can this installation capture RCCL collectives inside a hipGraph and replay them?  Tried where a hang costs nothing: every rank starts
    a CHILD process (same GPU, its own process group on its own port) that builds a small captioning step, captures it with
    capture(collectives=True), replays it and exits; a child that has not finished within the bound is killed.  Returns (ok on EVERY rank,
    reason).  The parents only exchange one port number before and one flag after."""
    import socket
    import subprocess
    import torch.distributed as dist
    timeout_s = timeout_s or args.trial_timeout
    port_t = torch.zeros(1, dtype=torch.int64, device=dev)
    if rank == 0:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port_t[0] = sk.getsockname()[1]
    dist.broadcast(port_t, src=0)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(int(port_t[0])), RANK=str(rank), LOCAL_RANK=str(local_rank),
               WORLD_SIZE=str(world))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in [k for k in env if k.startswith(("TORCHELASTIC_", "TORCH_NCCL_ASYNC", "GROUP_", "ROLE_"))]:
        env.pop(k)      # (under torchrun the rendezvous store is the AGENT's: TORCHELASTIC_USE_AGENT_STORE makes rank 0 a client of it -- the
                        # children have no agent, their rank 0 must host the store on the trial's port)
    cmd = [sys.executable, os.path.abspath(__file__), "--graph-overlap-trial", "--gpus", str(world)] + (["--dry-run"] if args.dry_run else [])
    ok, why = 0, "?"
    try:
        p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
        try:
            out, err = p.communicate(timeout=timeout_s)
            ok = int(p.returncode == 0 and "TRIAL OK" in out)
            why = "captured and replayed" if ok else f"child exit code {p.returncode}: {(err or out).strip()[-200:]}"
        except subprocess.TimeoutExpired:
            try:
                os.killpg(p.pid, 9)          # the child is its own session / process group: exactly the processes started here
            except OSError:
                p.kill()
            p.communicate()
            why = f"no result within {timeout_s} s (killed)"
    except OSError as exc:
        why = f"could not start the trial: {exc}"
    flag = torch.tensor([ok], dtype=torch.int64, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    allok = bool(int(flag[0]))
    return allok, (why if (allok or not ok) else "another rank's trial failed")


def graph_overlap_trial_child(args):
    """the child of graph_overlap_trial: a small captioning step over this rank's GPU and a process group of the children, captured WITH its
    collectives and replayed.  --dry-run: the same protocol over gloo without a GPU (tests/test_bench_launcher.py); BMT_TRIAL_HANG=1 makes it
    hang on purpose (the parent's time bound is what is tested)."""
    import torch.distributed as dist
    world, rank, local_rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("BMT_TRIAL_HANG") == "1":
        time.sleep(3600)
    if args.dry_run:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.ones(1)
        dist.all_reduce(t)
        assert int(t[0]) == world
        print("TRIAL OK", flush=True)
        os._exit(0)
    import contextlib
    import io
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from bmt_amd import synthetic as syn
    from bmt_amd.model.captioning_module import BiModalTransformer
    from bmt_amd.train import CaptioningTrainStep
    V, Tv, Ta, Tc, B = 1000, 64, 200, 12, 8
    cfg = syn.cfg_config0(dout_p=0.1)
    cfg.device = str(dev)
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        model = BiModalTransformer(cfg, syn.FakeTrainDataset(V, syn.make_glove(V, cfg.d_model_caps))).to(dev)
    batch = syn.make_cap_batch(cfg, B, Tv, Ta, Tc, V, seed=5 + rank)
    fs = {k: v.to(dev) for k, v in batch["feature_stacks"].items()}
    caps = batch["captions"].to(dev)
    step = CaptioningTrainStep(model, cfg, syn.PAD_IDX, data_parallel=True, static_grads=True, overlap=True, seed=77, bucket_bytes=1 << 20,
                               collective="allreduce" if args.dp_collective == "auto" else args.dp_collective)
    step.capture(fs, caps, warmup=1, collectives=True)
    for _ in range(3):
        loss, _ = step.replay()
    torch.cuda.synchronize()
    assert math.isfinite(float(loss))
    dist.barrier()
    print("TRIAL OK", flush=True)
    sys.stdout.flush()
    os._exit(0)


def kernel_timer_pass(step, inputs, timer, ops, n_steps, warm=2):
    """n_steps eagerly issued steps with HIP events around every launch of a kernel class (KernelTimer) and around every step; returns
    the median step time in ms.
      * on ONE stream: the timed region runs the encoder's audio and video chains on two streams (ops.fork_side_stream); an event pair
        around a launch would time whatever else shares the GPU with it;
      * ``warm`` untimed eager steps first: the timed region replayed hipGraphs out of their private memory pool, so the first eager step
        on this stream allocates its ~5 GB of temporaries from the driver (hipMalloc: tens of ms of host time that would sit between an
        event and its kernel);
      * the GPU is parked behind a spin kernel long enough for the host to issue every step ahead of it, so launches are queued back to
        back and no host time lands inside an interval."""
    import statistics
    timer.reset()
    enc_streams, ops.ENC_STREAMS = ops.ENC_STREAMS, 1
    try:
        for _ in range(warm):
            step(*inputs)
        torch.cuda.synchronize()
        timer.enabled = True
        torch.cuda._sleep(int(2.4e9 * 0.015 * n_steps))        # ~15 ms of shader cycles per step the host has to issue
        evs = []
        for _ in range(n_steps):
            timer.begin_step()
            s = torch.cuda.Event(enable_timing=True)
            s.record()
            step(*inputs)
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            evs.append((s, e))
        torch.cuda.synchronize()
    finally:
        timer.enabled = False
        ops.ENC_STREAMS = enc_streams
    return statistics.median(s.elapsed_time(e) for s, e in evs)


def self_launch(args):
    """`python bench.py --gpus N` without a torch.distributed environment: become the launcher (one rank per GPU, rendezvous on
    127.0.0.1) -- the ranks run this same file with the same arguments; rank 0 prints the JSON line."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    sys.stdout.flush()
    sys.stderr.flush()
    os.execvpe(cmd[0], cmd, env)


def dry_run(args, world, rank):
    """launcher / rendezvous / timing protocol without a GPU (gloo): what CI can check of `--gpus N` on a CPU-only box"""
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    trial_res = None
    if world > 1 and args.dp_mode == "auto":      # the child-process trial that guards the captured-allreduce mode: same protocol, gloo children
        trial_res = graph_overlap_trial(args, world, rank, rank, torch.device("cpu"))
    for _ in range(args.warmup):
        time.sleep(0.001)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001 * (rank + 1))
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt, 100.0 * (rank + 1)], dtype=torch.float64)
    if world > 1:
        tmax, tsum = t.clone(), t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt, units = float(tmax[0]), float(tsum[1])
    else:
        units = float(t[1])
    ranks_seen, rank_devices = 1, [{"rank": 0, "local_rank": 0, "device": "cpu"}]
    if world > 1:
        ones = torch.ones(1)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        ranks_seen = int(round(float(ones[0])))
        rank_devices = [None] * world
        dist.all_gather_object(rank_devices, {"rank": rank, "local_rank": rank, "device": "cpu"})
    if rank == 0:
        colls = ["allreduce", "rs_ag"] if args.dp_collective == "auto" else [args.dp_collective]
        flush_map = None
        if args.procedure == "train_cap":
            # which gradient bucket becomes final at which flush point of the overlapped backward pass (the configs[1] model on the CPU: the
            # bucket layout is pure arithmetic over the parameter list), so that a multi-GPU line can be checked against SURVEY.md 8e's budget
            # (exposed communication + imbalance <= 25 % of the step) from the line alone: bytes_after_last_layer_flush is what no backward
            # work is left to hide
            import contextlib, io
            from bmt_amd import ops as _ops, parallel as _par, synthetic as syn
            from bmt_amd.model.captioning_module import BiModalTransformer
            cfg = syn.cfg_config1(dout_p=0.1)
            cfg.device = "cpu"
            torch.manual_seed(0)
            with contextlib.redirect_stdout(io.StringIO()):
                model = BiModalTransformer(cfg, syn.FakeTrainDataset(10000, syn.make_glove(10000, cfg.d_model_caps)))
            flush_map = _par.bucket_flush_map(model, 32 << 20, _ops.fused_weight_groups(model))
        print(json.dumps({"metric": "dry run of the launcher (no GPU work)", "dry_run": True, "value": units * args.steps / dt, "unit": "units/s",
                          "dp_flush_map": flush_map,
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
                          "captured_allreduce_trial": trial_res, "rccl_ranks_seen": ranks_seen, "rank_devices": rank_devices,
                          "allreduce_exposed_ms": None, "dp_collective": colls[0], "dp_collectives_tried": colls if world > 1 else []}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def build_cap(args, dev, rank, world):
    from bmt_amd import ops, synthetic as syn
    from bmt_amd.model.captioning_module import BiModalTransformer
    from bmt_amd.train import CaptioningTrainStep
    V, Tv, Ta, Tc, B = 10000, 256, 800, 30, args.batch or 32
    cfg = syn.cfg_config1(dout_p=0.1)
    cfg.device = str(dev)
    torch.manual_seed(0)                                   # identical replicas on every rank
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        model = BiModalTransformer(cfg, syn.FakeTrainDataset(V, syn.make_glove(V, cfg.d_model_caps))).to(dev)
    n_params = sum(p.numel() for p in model.parameters() if p.requires_grad)
    # the timed region rotates NB pre-staged batches (round 6): with packed rows a step's time depends on its batch's raggedness, one batch
    # replayed K times would report that batch's.  Batch 0 is the one of the earlier rounds' lines (seed 1234 + rank).
    NB = max(1, args.batches)
    batches = [syn.make_cap_batch(cfg, B, Tv, Ta, Tc, V, seed=1234 + rank + 1000 * i) for i in range(NB)]
    batch = batches[0]
    staged = [({k: v.to(dev) for k, v in bt["feature_stacks"].items()}, bt["captions"].to(dev)) for bt in batches]      # inputs resident in HBM before timing
    fs, caps = staged[0]
    units_each = [int((c[:, 1:] != syn.PAD_IDX).sum()) for _, c in staged]
    units_local = units_each[0]
    step = CaptioningTrainStep(model, cfg, syn.PAD_IDX, data_parallel=world > 1, static_grads=True, seed=1000,
                               collective="allreduce" if args.dp_collective == "auto" else args.dp_collective)
    # how much of the padded batch is real: the encoder runs on the valid rows only (bmt_amd.ops.PACK_ROWS); every FLOP figure of this line
    # stays the PADDED-DENSE count of SURVEY.md 8d, so skipped padding reads as speed, never as skipped work
    va, vv = int(batch["La"].sum()), int(batch["Lv"].sum())
    fracs = [(int(bt["La"].sum()) + int(bt["Lv"].sum())) / float(B * (Ta + Tv)) for bt in batches]
    desc = {"batches": staged, "units_each": units_each, "valid_row_fractions": fracs,
            "flops_step_executed_each": [cap_step_flops(bt["La"].tolist(), bt["Lv"].tolist(), Ta, Tv, cfg.N) for bt in batches],
            "timer_valid": ({B * Ta: va, B * Tv: vv}, {Ta: batch["La"].tolist(), Tv: batch["Lv"].tolist()}),
            "metric": "caption tokens/sec/node (train_cap B=32/GPU, d=1024)", "unit": "caption tokens/s",
            "workload": "configs[1]: train_cap, N=2 d_model=1024 H=4 d_aud=128 d_vid=1024 d_caps=300 T_v=256 T_a=800 T_c=30 V=10000, "
                        "dropout 0.1, Adam, GloVe frozen",
            "B": B, "n_params": n_params, "units_name": "tokens_per_step",
            "valid_rows": {"audio": va, "audio_padded": B * Ta, "video": vv, "video_padded": B * Tv, "fraction": (va + vv) / float(B * (Ta + Tv)),
                           "packed": bool(ops.PACK_ROWS),
                           "note": "valid (non-padded) positions of batch 0 of this rank's synthetic batches (the kernel timer's batch); with packed rows the "
                                   "encoder's row-wise kernels and attention run over these only -- algorithmic_tflops / mfma_peak_frac / roofline.frac keep "
                                   "the padded-dense FLOP count of SURVEY.md 8d, the *_executed figures beside them count the rows that exist"},
            # algorithmic flops of the padded-dense step: 3.257 TFLOP per B=32 train step at V~10k
            "flops_step": 3.257e12 * (B / 32.0)}
    return step, (fs, caps), units_local, desc


def build_prop(args, dev, rank, world):
    from bmt_amd import ops, synthetic as syn
    from bmt_amd.model.proposal_generator import MultimodalProposalGenerator
    from bmt_amd.train import ProposalTrainStep
    B, Tv, Ta = args.batch or 16, 1024, 3200
    cfg = syn.cfg_config1(procedure="train_prop", dout_p=0.1, lr=1e-4)
    cfg.device, cfg.grad_clip = str(dev), None
    anchors = {"audio": syn.make_anchors(cfg.anchors_num_audio), "video": syn.make_anchors(cfg.anchors_num_video)}
    torch.manual_seed(0)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        model = MultimodalProposalGenerator(cfg, anchors).to(dev)
    for p in model.encoder.parameters():        # configs[3]: the bi-modal encoder comes from the captioning model and stays frozen
        p.requires_grad = False
    n_params = sum(p.numel() for p in model.parameters() if p.requires_grad)
    batch = syn.make_prop_batch(cfg, B, Tv, Ta, seed=11 + rank)
    fs = {k: v.to(dev) for k, v in batch["feature_stacks"].items()}
    tg = batch["targets"].to(dev)
    step = ProposalTrainStep(model, cfg, syn.PAD_IDX, data_parallel=world > 1, seed=1000, static_grads=True,
                             collective="allreduce" if args.dp_collective == "auto" else args.dp_collective)
    # heads 348.7 + 322.9 GF and encoder 230.0 GF forward per sample at (T_a, T_v) = (3200, 1024); the frozen
    # encoder has no backward, the heads have 2x their forward
    heads, enc = (348.7 + 322.9) * 1e9, 230.0e9
    desc = {"metric": "videos/sec/node (train_prop B=16/GPU, frozen bi-modal encoder, T_v=1024 T_a=3200)", "unit": "videos/s",
            "workload": "configs[3]: train_prop, bi-modal proposal generator (10+10 Conv1d heads, 48/128 anchors) over full-video streams "
                        "T_v=1024 T_a=3200, encoder of the configs[1] width frozen, dropout 0.1, Adam",
            "B": B, "n_params": n_params, "units_name": "videos_per_step", "flops_step": B * (enc + 3.0 * heads)}
    return step, (fs, tg), B, desc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--procedure", default="train_cap", choices=["train_cap", "train_prop"],
                    help="train_cap = BASELINE.json's metric (configs[1]); train_prop = configs[3]")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (weak scaling); default 32 (train_cap) / 16 (train_prop)")
    ap.add_argument("--batches", type=int, default=4, help="train_cap: distinct pre-staged synthetic batches the timed region rotates through")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--no-prefetch", action="store_true", help="train_cap, captured: copy each batch into the static input buffers in front of its own replay "
                    "instead of beside the previous step's optimizer graph")
    ap.add_argument("--timer-steps", type=int, default=7, help="eagerly issued steps of the per-kernel HIP-event pass (>= 5)")
    ap.add_argument("--no-clock-probe", action="store_true", help="skip the 2 s of back-to-back steps under rocm-smi after the timed region")
    ap.add_argument("--no-graph", action="store_true", help="issue every kernel from Python instead of replaying hipGraphs")
    ap.add_argument("--dp-mode", default="auto", choices=["auto", "graph", "overlap", "graph-overlap"],
                    help="N > 1: hipGraphs with the all-reduce exposed between them, eager launches with the all-reduce overlapped with "
                         "the backward pass, or (auto) whichever a short trial finds faster")
    ap.add_argument("--dry-run", action="store_true", help="launcher / rendezvous / timing protocol only (gloo, no GPU)")
    ap.add_argument("--dp-collective", default="auto", choices=["auto", "allreduce", "rs_ag"],
                    help="N > 1: a bucket's gradient sum as one all-reduce, as reduce-scatter + all-gather (SURVEY 5: world-1 simultaneous "
                         "point-to-point transfers per half on the xGMI mesh), or (auto) whichever the launch-mode trials find faster")
    ap.add_argument("--trial-timeout", type=int, default=240, help="seconds the child-process trial of the captured-allreduce mode may take")
    ap.add_argument("--graph-overlap-trial", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--threads", type=int, default=0, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_worker:
        return cpu_baseline_worker()
    if args.graph_overlap_trial:
        return graph_overlap_trial_child(args)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} (or without a "
                         "torch.distributed environment: bench.py starts its own ranks)")
    if args.dry_run:
        return dry_run(args, world, rank)

    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from bmt_amd import ops
    cap = args.procedure == "train_cap"
    step, inputs, units_local, desc = (build_cap if cap else build_prop)(args, dev, rank, world)
    B = desc["B"]

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

    # ---- train_cap: the step is captured into hipGraphs with the RCCL all-reduce of the flat gradient buckets issued between /
    # beside them (nothing collective is captured); eager issue of every kernel is the fallback.  train_prop: eager (the number
    # of target events changes per batch).
    mode = "eager"
    run = lambda: step(*inputs)
    run_with = lambda inp: step(*inp)          # the timed region's call: a step on the batch it is handed
    if hasattr(timer, "set_valid") and desc.get("timer_valid"):
        timer.set_valid(*desc["timer_valid"])

    def trial(fn, n=4):
        """ms per step of n steps, max over ranks (mode selection; outside the timed region)"""
        fn()
        sync()
        t_ = time.perf_counter()
        for _ in range(n):
            fn()
        sync()
        v = torch.tensor([(time.perf_counter() - t_) / n * 1e3], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(v, op=dist.ReduceOp.MAX)
        return float(v[0])

    mode_trials = {}
    if cap and not args.no_graph:
        # N == 1: the captured step.  N > 1 (--dp-mode auto) has three ways to run it; each is timed briefly (max over ranks) and the fastest
        # is what the timed region measures:
        #   eager+overlap                bucket all-reduces launched from the backward pass, every kernel issued from Python
        #   hipgraph                     two hipGraphs, the all-reduce of the flat buckets exposed between them
        #   hipgraph+captured-allreduce  ONE hipGraph with the bucket all-reduces captured inside it: overlap with no host work between
        #                                kernels.  A collective that cannot be captured on an installation may HANG instead of raising, so
        #                                this mode enters the comparison only after a trial in CHILD processes (one per rank, their own
        #                                process group) has captured and replayed it within a time bound (graph_overlap_trial).
        want = {"auto": ("eager", "graph", "graph-overlap"), "overlap": ("eager",), "graph": ("graph",),
                "graph-overlap": ("graph-overlap",)}[args.dp_mode] if world > 1 else ("graph",)
        if world > 1 and "graph-overlap" in want and args.dp_mode == "auto":
            ok, why = graph_overlap_trial(args, world, rank, local_rank, dev)
            note(f"captured-allreduce trial in child processes: {'ok' if ok else 'NOT usable'} ({why})")
            if not ok:
                want = tuple(m for m in want if m != "graph-overlap")
                mode_trials["hipgraph+captured-allreduce"] = f"not tried: {why}"
        # ... each with a bucket's sum as ONE all-reduce or as reduce-scatter + all-gather (--dp-collective auto: both are tried)
        colls = (("allreduce", "rs_ag") if args.dp_collective == "auto" else (args.dp_collective,)) if world > 1 else (None,)
        key = lambda name, c: name if len(colls) == 1 else f"{name}/{c}"

        def use(c):
            if c is not None and getattr(step, "reducer", None) is not None:
                step.reducer.set_collective(c)
        if "eager" in want:
            for c in colls:
                use(c)
                for _ in range(2):
                    step(*inputs)
                mode_trials[key("eager+overlap", c)] = trial(lambda: step(*inputs))
                note(f"{key('eager+overlap', c)}: {mode_trials[key('eager+overlap', c)]:.2f} ms/step")
        captured = None
        n_trials = len(want) * len(colls)
        for m in ("graph-overlap", "graph"):        # ("graph" last: when it wins or ties, its capture is the one that stays)
            if m not in want:
                continue
            for c in colls:
                name = key("hipgraph+captured-allreduce" if m == "graph-overlap" else "hipgraph", c)
                try:
                    use(c)
                    step.capture(*inputs, warmup=2, collectives=(m == "graph-overlap"))
                    captured = name
                    mode_trials[name] = trial(lambda: step.replay()) if n_trials > 1 else 0.0
                    if n_trials > 1:
                        note(f"{name}: {mode_trials[name]:.2f} ms/step")
                except Exception as exc:      # noqa: BLE001 -- e.g. a collective that refuses capture: keep measuring, say so
                    note(f"{name}: capture failed ({type(exc).__name__}: {exc})")
                    mode_trials[name] = f"capture failed: {type(exc).__name__}"
                    torch.cuda.synchronize()
                    step.uncapture()
                    captured = None
        timed = {k: v for k, v in mode_trials.items() if isinstance(v, float)}
        best = min(timed, key=timed.get) if timed else None
        best_mode, _, best_coll = (best or "").partition("/")
        use(best_coll or (colls[0] if colls[0] is not None else None))
        if best is None or best_mode == "eager+overlap":
            step.uncapture()
        else:
            if captured != best:          # the winner's graphs were replaced by a later capture: capture it again
                step.capture(*inputs, warmup=1, collectives=(best_mode == "hipgraph+captured-allreduce"))
            run = lambda: step.replay()
            # (the batch is copied into the graphs' static input buffers; the NEXT step's batch is announced with the call, so that its copy runs
            # beside this step's optimizer graph: CaptioningTrainStep.replay)
            run_with = lambda inp, nxt=None: step.replay(*inp, next_batch=nxt)
            mode = best_mode
        if mode == "eager" and world > 1:
            mode = "eager+overlap"
    if (not cap) and not args.no_graph and world == 1:
        # train_prop on one GPU: the whole step (zero_grad .. Adam) as one hipGraph over a batch padded to a fixed number of target rows
        # (ProposalTrainStep.capture); N > 1 launches eagerly (the obj / noobj counts and the gradients are all-reduced from inside the step)
        try:
            step.capture(*inputs, warmup=2)
            run = lambda: step.replay()
            run_with = lambda inp: step.replay()
            mode = "hipgraph"
        except Exception as exc:      # noqa: BLE001
            note(f"train_prop: graph capture failed ({type(exc).__name__}: {exc}); eager launches")
            torch.cuda.synchronize()
    rot = desc.get("batches") or [inputs]          # train_cap: NB pre-staged batches, handed to the step in turn
    prefetch = cap and mode.startswith("hipgraph") and len(rot) > 1 and not args.no_prefetch
    for i in range(args.warmup):
        res = run_with(rot[i % len(rot)])
    sync()
    note(f"warmup done ({args.warmup} steps, {mode}, {len(rot)} batch(es) in rotation)")
    if hasattr(step, "reduce_timing"):
        step.reduce_timing(True)
    t0 = time.perf_counter()
    for i in range(args.steps):
        if prefetch:       # (the last step announces a batch too: every timed step carries one input copy, as without the prefetch)
            res = run_with(rot[i % len(rot)], rot[(i + 1) % len(rot)])
        else:
            res = run_with(rot[i % len(rot)])
    sync()
    dt = time.perf_counter() - t0
    if desc.get("units_each"):                     # the units the timed steps processed: each step's own batch
        units_local = sum(desc["units_each"][i % len(rot)] for i in range(args.steps)) / float(args.steps)
    exposed_ms = step.reduce_timing(False) if hasattr(step, "reduce_timing") else None
    final_loss = float(res[0] if cap else res[1])
    note(f"timed region done: {dt / args.steps * 1e3:.2f} ms/step ({mode})")
    # (the same launches as the timed region, just more of them: nothing about the step's state changes before the kernel timer below)
    clock = None
    if world == 1 and not args.no_clock_probe:
        clock = clock_under_load(run, sync)
        note(f"engine clock under load: {clock}")
    timer_steps, eager_ms, gate_notes, timer_passes = 0, None, [], 0
    if not args.no_kernel_timer:
        # per-kernel HIP-event timing needs individual launches: the same step, eagerly issued on ONE stream, right after the timed region;
        # a pass that fails its sanity gates (roofline_gates) is repeated once, and a second failure is REPORTED (roofline.valid false)
        timer_steps = max(5, args.timer_steps)
        for timer_passes in (1, 2):
            eager_ms = kernel_timer_pass(step, inputs, timer, ops, timer_steps, warm=2 if timer_passes == 1 else 1)
            summ, used = timer.summary()
            gate_notes = roofline_gates({k: v["ms"] for k, v in summ.items()}, eager_ms, dt / args.steps * 1e3)
            if used < timer_steps:
                gate_notes.append(f"only {used} of {timer_steps} eager steps had the same launch sequence")
            note(f"kernel timer pass {timer_passes}: eager one-stream step {eager_ms:.2f} ms, classes {sum(v['ms'] for v in summ.values()):.2f} ms/step"
                 + ("" if not gate_notes else " -- GATES: " + "; ".join(gate_notes)))
            if not gate_notes:
                break

    ranks_seen, rank_devices = 1, None
    if world > 1:          # who took part, as the collective library and the ranks themselves report it
        ones = torch.ones(1, dtype=torch.float32, device=dev)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        ranks_seen = int(round(float(ones[0])))
        props = torch.cuda.get_device_properties(dev)
        mine = {"rank": rank, "local_rank": local_rank, "device": torch.cuda.current_device(), "name": props.name,
                "pci_bus_id": getattr(props, "pci_bus_id", None), "uuid": str(getattr(props, "uuid", "")) or None}
        rank_devices = [None] * world
        dist.all_gather_object(rank_devices, mine)
    t = torch.tensor([dt, float(units_local)], dtype=torch.float64, device=dev)
    if world > 1:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt, units_all = float(tmax[0]), float(tsum[1])
    else:
        units_all = float(units_local)

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = units_all * args.steps / dt
        flops_step = desc["flops_step"] * world
        out = {
            "metric": desc["metric"], "value": value, "unit": desc["unit"],
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "launch_mode": mode,
            "dtype": ops.precision_description(args.procedure), "data": "synthetic",
            "config": {"workload": desc["workload"], "global_batch": B * world, "per_gpu_batch": B, "parallelism": f"dp{world}",
                       "trainable_params": desc["n_params"], desc["units_name"]: units_all, "final_loss": final_loss},
            "valid_row_fraction": (desc.get("valid_rows") or {}).get("fraction"), "valid_rows": desc.get("valid_rows"),
            "algorithmic_tflops": flops_step / (ms_per_step * 1e-3) / 1e12,
            "mfma_peak_frac": flops_step / (ms_per_step * 1e-3) / 1e12 / (MFMA_BF16_DENSE_PEAK_TFLOPS * world),
        }
        if desc.get("flops_step_executed_each"):
            # what the hardware did (round 6): the FLOPs of the rows that exist, averaged over the batches the timed steps ran (rank 0's batches
            # stand for every rank's: same generator, same length distribution)
            fx = desc["flops_step_executed_each"]
            nb_rot = len(fx)
            fx_mean = sum(fx[i % nb_rot] for i in range(args.steps)) / float(args.steps) * world
            vf = desc["valid_row_fractions"]
            out["valid_row_fraction"] = sum(vf[i % nb_rot] for i in range(args.steps)) / float(args.steps)
            out["valid_row_fraction_per_batch"] = vf
            out["executed_tflops"] = fx_mean / (ms_per_step * 1e-3) / 1e12
            out["frac_executed"] = out["executed_tflops"] / (MFMA_BF16_DENSE_PEAK_TFLOPS * world)
            out["timed_region"] = (f"{args.steps} steps over {nb_rot} pre-staged batches in rotation, each handed to the step as device tensors "
                                   "(captured mode: copied into the graphs' static input buffers, ~80 MB per step, inside the timed region"
                                   + ("; each step announces the next step's batch, whose copy runs on a copy stream beside that step's optimizer "
                                      "graph -- a data loader's prefetch" if prefetch else "") + "); the loss stays on the "
                                   "device and is read once after the last step (the reference reads loss.item() every step: "
                                   "epoch_loops/captioning_epoch_loops.py:143)")
        if clock is not None:
            out["engine_clock_under_load"] = clock
        if world > 1:
            out["allreduce_exposed_ms"] = exposed_ms
            out["launch_mode_trials_ms"] = mode_trials
            out["allreduce"] = getattr(step, "reduce_description", lambda: None)()
            out["dp_collective"] = getattr(getattr(step, "reducer", None), "collective", None)
            out["rccl_ranks_seen"] = ranks_seen          # sum all-reduce of ones over the process group: what RCCL itself counts
            out["rank_devices"] = rank_devices
        if not args.no_kernel_timer:
            summ, used_steps = timer.summary()         # per class and STEP: launches, ms (sum of per-launch medians), flops, bytes
            valid = not gate_notes
            tot = sum(v["ms"] for v in summ.values()) or 1.0
            cand = {}
            for k, v in summ.items():          # the attention backward competes as ONE class (encoder- and decoder-sized launches together)
                kk = "attn_bwd (encoder + decoder launches)" if k.startswith("attn_bwd_") else k
                c = cand.setdefault(kk, {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "launches": 0, "ms_sum_of_means": 0.0, "ms_max_step": 0.0,
                                         "xflops": 0.0, "xbytes": 0.0})
                for f in c:
                    c[f] += v[f]
                if kk != k:
                    timer.passes.setdefault(kk, 1)
            dom = max(cand, key=lambda k: cand[k]["ms"])
            d = cand[dom]
            ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
            xach = d["xflops"] / (d["ms"] * 1e-3) / 1e12           # over the rows that exist (the kernel timer's batch: batch 0)
            rec, rec_note = pmc_record(args.procedure)
            fam = [(rec or {}).get("kernels", {}).get(k) for k in pmc_keys_of_class(dom)]
            kern = None
            if fam and all(fam):        # per launch of the class: the sum over its kernels (families mix encoder and decoder sizes)
                kern = {"traffic_bytes": sum(x["traffic_bytes"] for x in fam)}
                if all(x.get("mfma_busy") is not None and x.get("avg_us") for x in fam):
                    kern["mfma_busy"] = sum(x["mfma_busy"] * x["avg_us"] for x in fam) / sum(x["avg_us"] for x in fam)
                    kern["hbm_gbs"] = kern["traffic_bytes"] / sum(x["avg_us"] for x in fam) / 1e3
            passes = timer.passes.get(dom, 1)
            timing = (f"HIP events around every launch of the class (a split-K GEMM launch = main kernel + its epilogue kernel) in {used_steps} eagerly "
                      f"issued one-stream steps right after the timed region; per launch the MEDIAN over the steps, summed over the class's launches "
                      f"(kernel timer pass {timer_passes})")
            out["roofline"] = {"valid": valid, "kernel": dom, "bound": "mfma", "achieved": ach if valid else None, "peak": MFMA_BF16_DENSE_PEAK_TFLOPS,
                               "unit": "TFLOP/s", "frac": ach / MFMA_BF16_DENSE_PEAK_TFLOPS if valid else None,
                               "executed": xach if valid else None, "frac_executed": xach / MFMA_BF16_DENSE_PEAK_TFLOPS if valid else None,
                               # issued = what the matrix pipe executes: the FLOPs of the rows that exist x the MFMA passes of the operand format
                               "frac_issued": xach * passes / MFMA_BF16_DENSE_PEAK_TFLOPS if valid else None,
                               "traffic": kern["traffic_bytes"] if kern else None,
                               "traffic_unit": "HBM bytes per launch", "traffic_source": rec_note if kern or rec is None else rec_note + f" -- no entry for {dom}",
                               "algorithmic_bytes_per_launch": d["xbytes"] / d["launches"],
                               "algorithmic_bytes_per_launch_padded": d["bytes"] / d["launches"],
                               "traffic_over_algorithmic": (kern["traffic_bytes"] / (d["xbytes"] / d["launches"])) if kern else None,
                               "flop_conventions": "achieved / frac: padded-dense FLOPs (SURVEY.md 8d); executed / frac_executed / frac_issued and "
                                                   "algorithmic_bytes_per_launch: the rows that exist in the kernel timer's batch (packed rows)",
                               "launches": d["launches"], "avg_launch_us": d["ms"] * 1e3 / d["launches"], "ms_per_step": d["ms"],
                               "ms_per_step_mean_of_intervals": d["ms_sum_of_means"], "ms_worst_step": d["ms_max_step"],
                               "share_of_timed_kernels": d["ms"] / tot, "mfma_passes": passes, "timing": timing}
            if not valid:          # the numbers of an inconsistent pass are shown for diagnosis under another name, never as the roofline
                out["roofline"]["invalid_because"] = gate_notes
                out["roofline"]["rejected_achieved"] = ach
            if kern and kern.get("mfma_busy") is not None:
                out["roofline"]["mfma_busy"] = kern["mfma_busy"]
                out["roofline"]["hbm_gbs"] = kern["hbm_gbs"]
                out["roofline"]["pmc_families"] = list(pmc_keys_of_class(dom))
            # the north-star quantity: bi-modal ENCODER attention against the MFMA roofline.  "issued" counts what the matrix
            # pipe executes (backward: 7 products as scheduled -- S and dP are computed by both backward kernels),
            # "algorithmic" the 2 / 5 products of the math.
            enc = {k: v for k, v in summ.items() if k.startswith(("attn_fwd_enc", "attn_bwd_enc"))}
            if enc:
                ms = sum(v["ms"] for v in enc.values())
                alg = sum(v["flops"] for v in enc.values())
                alg2 = sum(v["flops"] * (1.0 if k.startswith("attn_fwd") else 0.8) for k, v in enc.items())
                xalg = sum(v["xflops"] for v in enc.values())            # sum_b Lq_b Lk_b instead of B Sq Sk
                # (the two-kernel and the recompute backward compute S and dP on both sides: 7 products for the 5 of the math; the emitting
                # split form issues the 5) -- over the rows that exist
                issued = sum(v["xflops"] * (timer.passes.get(k, 1) if k.startswith("attn_fwd") else (1.0 if k.endswith("_split") else 1.4))
                             for k, v in enc.items())
                out["attention_roofline"] = {
                    "valid": valid,
                    "scope": "encoder self- and cross-attention cores, forward + backward, B=32 H=4 d_k=256 T_v=256 T_a=800",
                    "bound": "mfma", "peak": MFMA_BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "algorithmic": alg / (ms * 1e-3) / 1e12, "issued": issued / (ms * 1e-3) / 1e12,
                    "frac_algorithmic": alg / (ms * 1e-3) / 1e12 / MFMA_BF16_DENSE_PEAK_TFLOPS,
                    "executed": xalg / (ms * 1e-3) / 1e12, "frac_executed": xalg / (ms * 1e-3) / 1e12 / MFMA_BF16_DENSE_PEAK_TFLOPS,
                    # the same time against SURVEY.md 8d's own count (backward = 2 x forward: 4 products, the recomputed S = Q K^T not counted)
                    "frac_algorithmic_bwd_2x_fwd": alg2 / (ms * 1e-3) / 1e12 / MFMA_BF16_DENSE_PEAK_TFLOPS,
                    "rank_form": "the audio self-attention and the video stream's attention over the audio stream run at the key / value input's width "
                                 "(128) against that input as the heads' shared key / value plane (classes *_rank_of_dk256): algorithmic = the reference's "
                                 "formulation at d_k 256, executed / issued = the products the kernels run (half the width)",
                    "flop_conventions": "frac_algorithmic: forward 2 products + backward 5 (incl. the recomputed scores) over PADDED (B, S, S) -- "
                                        "padding is not computed under packed rows; frac_algorithmic_bwd_2x_fwd: backward counted as 4 products; "
                                        "executed / frac_executed: the same 2 + 5 products over sum_b Lq_b Lk_b of the kernel timer's batch; "
                                        "issued / frac_issued: executed x (forward passes | backward products run / 5)",
                    "frac_issued": issued / (ms * 1e-3) / 1e12 / MFMA_BF16_DENSE_PEAK_TFLOPS,
                    "ms_per_step": ms, "ms_per_step_forward": sum(v["ms"] for k, v in enc.items() if k.startswith("attn_fwd")),
                    "ms_per_step_backward": sum(v["ms"] for k, v in enc.items() if k.startswith("attn_bwd")),
                    "share_of_timed_kernels": ms / tot}
                if rec:          # rocprofv3 --pmc passes over the same step (tools/gpu_pmc_bench.sh), per encoder attention kernel
                    pm = {k: {x: v.get(x) for x in ("mfma_busy", "hbm_gbs", "launches_seen", "avg_us")} for k, v in rec["kernels"].items()
                          if k.startswith("attn_") and v.get("mfma_busy") is not None}
                    if pm:
                        out["attention_roofline"]["pmc"] = pm
                        out["attention_roofline"]["pmc_source"] = rec_note
            out["kernel_timer"] = {"valid": valid, "gates": gate_notes, "passes_run": timer_passes, "steps": used_steps,
                                   "eager_ms_per_step": eager_ms, "streams": 1, "timed_region_streams": ops.encoder_streams_in_use(),
                                   "timed_classes_ms_per_step": tot,
                                   "note": "HIP events on torch's current stream around every launch of a class and around every step; two untimed "
                                           "eager steps first (allocator), then the GPU is parked behind a spin kernel so that launches are queued "
                                           "back to back; gates: classes <= eager step, eager step <= 1.6 x ms_per_step, every class <= ms_per_step"}
            out["kernel_classes"] = {k: {"ms_per_step": v["ms"], "tflops": v["flops"] / (v["ms"] * 1e-3) / 1e12,
                                         "tflops_executed": v["xflops"] / (v["ms"] * 1e-3) / 1e12,
                                         "gbs": v["bytes"] / (v["ms"] * 1e-3) / 1e9, "launches_per_step": v["launches"],
                                         "ms_per_step_mean_of_intervals": v["ms_sum_of_means"]}
                                     for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])}
        if world == 1 and cap and not args.no_cpu_baseline:
            note("timing the CPU oracle (bounded sample, child process)")
            out["cpu_baseline"] = cpu_baseline()
        elif world == 1 and not cap:
            out["cpu_baseline"] = {"value": None, "unit": desc["unit"], "cores": None, "kind": "port",
                                   "sample": "not timed: the reference at this size needs >10 min per step on the host (BASELINE.md 2: 6.3 s "
                                             "for B=2 at T_v=300/T_a=800); the train_cap line carries the CPU baseline"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
        # the line is out and every rank is done: leave without destroy_process_group and without the interpreter's teardown (on this
        # image destroy_process_group aborts now and then -- SIGABRT from a c10d / RCCL watchdog thread after a good run)
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)


if __name__ == "__main__":
    main()
