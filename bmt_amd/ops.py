"""Host-side operators: thin wrappers over the C ABI plus the ``torch.autograd.Function``s the
model classes are built from.  PyTorch is plumbing here (device memory, streams, autograd
graph); every FLOP on the path is issued by libbmt_hip.so.

Precision policy (DESIGN.md "precision", tests/study_precision_policy.py): every forward product names the MFMA operand
format of its site -- the encoder's GEMMs and the decoder's memory K/V projections run A(fp16) x W(fp16 hi + lo), two passes;
every attention core runs single-pass fp16; the decoder's own small GEMMs, the bridge and the generator run split-bf16 (three
passes) -- chosen so that the log-probabilities stay within 1e-3 of the fp32 reference with >= 2x margin.  Backward products
run single-pass bf16.
"""
from __future__ import annotations

import ctypes as C
import math
import os as _os
from typing import Optional

import torch

from . import _lib
from ._lib import (EPI_ACCUM, EPI_BIAS, EPI_DROP_POST, EPI_DROP_PRE, EPI_GATE, EPI_RELU, EPI_RESIDUAL, PREC_BF16,
                   PREC_BF16X3, PREC_F16, PREC_F16W2, AttnBwdArgs, AttnBwdBf16Args, AttnFwdArgs, AttnFwdBf16Args, GemmBf16Args)

lib = _lib.load()

# ----------------------------------------------------------------------------- precision policy
BWD_PRECISION = PREC_BF16
_PREC = {PREC_BF16: ("bf16", 1, (2.0, 2.0)), PREC_BF16X3: ("bf16x3", 3, (4.0, 4.0)),
         PREC_F16: ("fp16", 1, (2.0, 2.0)), PREC_F16W2: ("fp16 x (fp16 hi+lo)", 2, (2.0, 4.0))}


def prec_name(prec: int) -> str:
    return _PREC[prec][0]


def prec_passes(prec: int) -> int:
    """MFMA passes issued per algorithmic product"""
    return _PREC[prec][1]


def prec_operand_bytes(prec: int):
    """(A, B) operand plane bytes per element read by a product of this precision"""
    return _PREC[prec][2]


class Policy:
    """forward operand formats of one module family: ``gemm`` for its projections / FFN, ``kv_gemm`` for the key / value projections
    of a CROSS-attention (their input is the long encoder memory), ``attn`` for the attention core"""
    __slots__ = ("gemm", "kv_gemm", "attn", "name", "ffn2", "ffn1")

    def __init__(self, gemm, kv_gemm, attn, name, ffn2=None, ffn1=None):
        self.gemm, self.kv_gemm, self.attn, self.name = gemm, kv_gemm, attn, name
        self.ffn2 = gemm if ffn2 is None else ffn2      # the second product of a PositionwiseFeedForward
        self.ffn1 = gemm if ffn1 is None else ffn1      # the first


# max |d log-prob| vs the fp32 reference on the mid fixture with this table: 3.7e-4 (CPU emulation, tests/study_precision_policy.py;
# one site at a time moved from two-plane to one-plane fp16 on top of it: FFN-2 3.8e-4, FFN-1 4.7e-4, Q 4.9e-4, K/V 5.8e-4,
# out-projection 7.2e-4, all encoder GEMMs 1.0e-3).  FFN-2 on one plane looks free there (-0.15 ms / step, 10.72 ms) but the six
# encoder layers of configs[4] accumulate it: the deep proposal generator's predictions leave their 1e-3 bar
# (test_deep_config_proposal_generator) -- not taken.
POLICIES = {
    "enc": Policy(PREC_F16W2, PREC_F16W2, PREC_F16, "enc"),       # bi-modal encoder layers (89 % of the FLOPs)
    # ... of an encoder of at most two layers (configs[0], [1], [3]: model/encoders.py tags by depth): FFN-2 on one fp16 plane.  The rounding
    # of the weight it admits accumulates with depth -- fine at N = 2 (log-probs 3.8e-4 against 3.7e-4 on the mid fixture), over the bar at the
    # six layers of configs[4] (see above), which keep two planes everywhere
    "enc_shallow": Policy(PREC_F16W2, PREC_F16W2, PREC_F16, "enc_shallow", ffn2=PREC_F16),
    # bi-modal decoder layers: their own GEMMs split-bf16 (fp16 x split-fp16 there leaves the 1e-3 bar: 9.3e-4 / 9.4e-4 on the mid fixture
    # and configs[0], round 4), the key / value projections of the long encoder memories as the encoder's
    "dec": Policy(PREC_BF16X3, PREC_F16W2, PREC_F16, "dec"),
    # Conv1d stacks of the proposal heads: split-bf16.  (fp16 activation x split weight leaves 6-8e-4 abs on the head outputs,
    # tests/test_gpu_proposal.py at round 2: inside the 1e-3 bar but with < 2x margin, and exp() turns it into 1e-3 relative on
    # the predicted lengths.)
    "head": Policy(PREC_BF16X3, PREC_BF16X3, PREC_BF16X3, "head"),
    # ... except a head's FIRST layer, the k-tap Conv1d (k up to 211 taps over 1024 channels: 90 % of a head's FLOPs, the dominant class of
    # train_prop): ONE fp16 pass (round 6; rounds 4-5: fp16 activation x split fp16 weight, two passes).  The study
    # (tools/probes/head_conv_one_pass.py, profiles/r06_v_head_conv_one_pass.txt): at the reference's real sizes a head's outputs are
    # 3.2e-5 from the reference with one pass, 2.0e-5 with two, 9e-7 with three bf16 passes (bar 1e-3; |y| <= 0.13 at initialisation) -- the
    # activation's own fp16 rounding is in both, so the second weight plane bought a factor sqrt 2, not an order of magnitude; a power-of-two
    # scale on the weights (1-2 % of them are fp16 subnormals) changes nothing measurable.  Every model-level proposal fixture holds its bar
    # unchanged; train_prop 49.2 -> 41.7 ms/step.  The 1 x 1 layers behind it -- directly under the sigmoid / exp of the predictions -- stay
    # split-bf16 (model/proposal_generator.py tags the heads by the encoder's depth: under a deep encoder the k-tap layer keeps three passes)
    "head_conv": Policy(PREC_F16, PREC_BF16X3, PREC_BF16X3, "head_conv"),
    None: Policy(PREC_BF16X3, PREC_BF16X3, PREC_BF16X3, "x3"),    # everything else: bridge, generator, embedders, uni-modal models
}
_OVERRIDE = [None]      # a Policy applied to EVERY site (A/B measurements, tests), or None


def set_precision(fwd: Optional[int] = None, bwd: int = PREC_BF16):
    """fwd = None: the per-site policy table (default).  fwd = PREC_*: that format at every forward site (PREC_F16W2 for GEMMs
    implies PREC_F16 attention cores)."""
    global BWD_PRECISION
    BWD_PRECISION = bwd
    if fwd is None:
        _OVERRIDE[0] = None
    else:
        attn = PREC_F16 if fwd == PREC_F16W2 else fwd
        _OVERRIDE[0] = Policy(fwd, fwd, attn, prec_name(fwd))


def policy_of(module_or_tag) -> Policy:
    if _OVERRIDE[0] is not None:
        return _OVERRIDE[0]
    tag = module_or_tag if (module_or_tag is None or isinstance(module_or_tag, str)) else getattr(module_or_tag, "bmt_policy", None)
    return POLICIES[tag]


def tag_policy(module: torch.nn.Module, tag: Optional[str]):
    """mark a module tree (an encoder / decoder layer) with the policy its sites run under"""
    for m in module.modules():
        m.bmt_policy = tag
    return module


def precision_description(procedure: Optional[str] = None) -> str:
    if procedure == "train_prop":
        hc, hd = POLICIES["head_conv"], POLICIES["head"]
        return (precision_description() + f"; proposal heads: the k-tap Conv1d {prec_name(hc.gemm)} ({prec_passes(hc.gemm)} pass; {prec_name(hd.gemm)} under an "
                f"encoder of more than two layers), the 1 x 1 layers {prec_name(hd.gemm)} ({prec_passes(hd.gemm)} passes)")
    if _OVERRIDE[0] is not None:
        o = _OVERRIDE[0]
        return f"every forward GEMM {prec_name(o.gemm)}, attention forward {prec_name(o.attn)}, backward {prec_name(BWD_PRECISION)} MFMA operands; fp32 accumulate"
    e, d, x, sh = POLICIES["enc"], POLICIES["dec"], POLICIES[None], POLICIES["enc_shallow"]
    ffn2 = "" if sh.ffn2 == sh.gemm else f"; FFN-2 of an encoder of <= 2 layers {prec_name(sh.ffn2)}, {prec_passes(sh.ffn2)} pass"
    raw = globals().get("RAW_MEMORY", False)
    mem = (f"the decoder's cross-attentions run against the raw encoder memories (no K/V projections: per-head block products {prec_name(x.gemm)}, "
           f"the two products against the memory {prec_name(e.attn)})" if raw else f"decoder memory K/V projections {prec_name(e.gemm)}")
    rank = ""
    if globals().get("RANK_ATTN", False):
        rank = (" -- the attentions over the 128-wide audio stream (its self-attention" + (", the video stream's attention over it" if globals().get("RANK_CROSS", False) else "") +
                f") in the rank form: queries through W' = W_k^T W_q ({prec_name(e.gemm)}, built in fp32 once per step), keys = values = the stream's own fp16 plane")
    return (f"MFMA operands per site: encoder GEMMs {prec_name(e.gemm)} ({prec_passes(e.gemm)} passes{ffn2}), {mem}, "
            f"attention forward {prec_name(e.attn)} (1 pass){rank}, decoder GEMMs / bridge / generator {prec_name(x.gemm)} (3 passes), backward "
            f"{prec_name(BWD_PRECISION)} (1 pass; the encoder's attention backward on fp16 q / k / v with power-of-two scaled fp16 gradients); "
            f"fp32 accumulate, softmax, LayerNorm, loss, Adam")


WEIGHT_EPOCH = [0]      # bumped by the optimizer: invalidates cached weight planes


def weights_changed(*_):
    """call after writing parameters behind the optimizer's back (load_state_dict does, through a module hook)"""
    WEIGHT_EPOCH[0] += 1


# ----------------------------------------------------------------------------- plumbing
def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _f32c(t: torch.Tensor) -> torch.Tensor:
    """fp32, CUDA, last dim contiguous 2-D-viewable tensor."""
    if not t.is_cuda:
        raise RuntimeError("bmt_amd ops need CUDA/HIP tensors: there is no CPU fallback (use the oracle for CPU checks)")
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


class StepContext:
    """state that belongs to ONE forward / backward pass in flight: keyed by (device, stream), so two models stepping from two
    threads on two streams (or nn.DataParallel replicas on their devices) do not see each other's -- and found again from
    autograd's backward threads, which run a node on the stream its forward ran on."""
    __slots__ = ("defer_dw", "pending_dw", "pending_ids", "pending_done", "pending_cs", "pending_post", "pending_side", "res_offer", "last_ln", "kv_cache", "allow_streams", "last_gen", "gen_handles",
                 "step_start", "planes_ready", "planes_waited", "deferred", "deferred_join")

    def __init__(self):
        self.defer_dw = False        # queue the weight-gradient products of this backward pass for grouped launches (flush_dw)
        self.pending_dw = []
        self.pending_ids = set()     # parameters whose gradient product is queued: their "gradient final" report waits for the flush
        self.pending_done = []
        self.pending_post = []       # launches that read a queued product's result: run by flush_dw behind the grouped launch
        self.pending_side = []       # (product item, follow-up launch) pairs that run BESIDE the grouped launch, on the auxiliary stream: the rank-form
                                     # attentions' dW' = dq'^T y and the chain rule through W' behind it -- small, serial, and everything waits for them
        self.pending_cs = []         # queued column-sum reductions (LayerNorm dgamma / dbeta partials, attention bias partials): colsum_multi
        self.res_offer = None        # residual offered by a ResidualConnection to its sublayer's last GEMM
        self.last_ln = None          # operand planes written by the LayerNorm kernel that just ran
        self.kv_cache = None         # dict while bmt_amd.decode.greedy_decoder runs: id(attention module) -> (memory, k planes, v planes)
        self.allow_streams = True    # cleared by a train step whose gradient reducer needs autograd-order completion on ONE stream
        self.last_gen = None         # GenHandle of the GeneratorFn.forward that just ran (picked up by model.generators.Generator)
        self.gen_handles = []        # handles whose loss took the fused backward: flush_dw settles their parameters' use counts
        self.step_start = None       # event at the beginning of the pass in flight (mark_step_start): what the table stream of a grouped launch waits for
        self.planes_ready = None     # event behind the weight-plane refresh that mark_step_start issued on the refresh stream (EARLY_REFRESH), or None
        self.planes_waited = set()   # handles of the streams that wait for it already
        self.deferred = None         # a launch of this pass that only has to happen before its backward (defer_beside): the gradient arena's zero fill
        self.deferred_join = None    # the stream it was issued on (join_deferred waits for it)


_contexts = {}
_contexts_lock = __import__("threading").Lock()


_ctx_alias = {}           # (device, side stream) -> (device, main stream): a forked branch of ONE pass shares the pass's context


def context() -> StepContext:
    key = (torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream)
    key = _ctx_alias.get(key, key)
    c = _contexts.get(key)
    if c is None:
        with _contexts_lock:
            c = _contexts.setdefault(key, StepContext())
    return c


# ---- two compute streams for the bi-modal encoder (model/encoders.py): the audio and the video stream of a layer only meet at the
# cross-modal attention, so the video chain is issued on a side stream (forked from / joined into the stream the model runs on, also
# under hipGraph capture, where the fork becomes parallel branches of the graph).  What is per pass stays per pass: the side stream
# is an alias of the main stream's StepContext (queued weight gradients and small reductions are flushed once, on the main stream,
# after autograd has joined the streams); what is per stream is already keyed by stream (split-K and grouped-GEMM scratch).
ENC_STREAMS = int(_os.environ.get("BMT_ENC_STREAMS", "3"))     # switch: 1 = everything on one stream (the profilers' and the kernel timer's eager steps);
                                                               # 2 = without the first decoder layer's self-attention sublayer beside the encoder (round 5: 7.17 / 7.21 -> 7.13 / 7.19 ms)
_side_streams = {}


def side_stream(index: int = 0, main=None) -> "torch.cuda.Stream":
    """side stream ``index`` of a main stream (default: the current one): every main stream has its own, so that two models stepping
    from two threads on two streams of one device do not meet on a shared side stream"""
    main = torch.cuda.current_stream() if main is None else main
    key = (main.device.index, main.cuda_stream, index)
    s = _side_streams.get(key)
    if s is None:
        s = _side_streams[key] = torch.cuda.Stream(device=main.device)
    return s


def fork_side_stream(index: int = 0):
    """side stream ordered after everything issued so far on the current stream, sharing its StepContext; None when two streams are
    switched off.  The weight planes are refreshed first: a branch must not find them half-way through the once-per-step refresh that
    the other branch's first GEMM triggered.  index 0: the encoder's video chain / a decoder layer's video attention; 1: the decoder's
    first self-attention sublayer, which does not depend on the encoder (model/captioning_module.py)."""
    if ENC_STREAMS < 2 + index or not context().allow_streams:
        return None
    with _weights.lock:
        _weights.ensure_fresh()
    main = torch.cuda.current_stream()
    dev = torch.cuda.current_device()
    mkey = _ctx_alias.get((dev, main.cuda_stream))
    if mkey is not None:          # called on a side stream (a decoder layer inside a forked branch): no nested fork
        return None
    s2 = side_stream(index, main)
    _ctx_alias[(dev, s2.cuda_stream)] = (dev, main.cuda_stream)
    s2.wait_stream(main)
    return s2


def join_side_stream(release: bool = True):
    """the current stream waits for its side streams: call after a backward pass whose forward forked (autograd runs a node on the stream its
    forward ran on and orders streams along gradient edges only -- the last nodes of the side chain write static gradient buffers and queue
    weight-gradient operands without handing anything to a node of the main stream).  ``release``: the pass is over -- its side streams stop
    being aliases of this stream's StepContext (torch hands out stream handles from a pool of 32 per priority: an alias that outlived its
    pass could make a LATER main stream with the same handle resolve to this pass's context)."""
    cur = torch.cuda.current_stream()
    for (d, m, _), s2 in list(_side_streams.items()):
        if d == cur.device.index and m == cur.cuda_stream:
            cur.wait_stream(s2)
    if release:
        release_side_streams(cur)


def release_side_streams(main=None):
    """forget the context aliases of ``main``'s side streams (the stream objects stay cached for the next pass)"""
    main = torch.cuda.current_stream() if main is None else main
    mkey = (main.device.index, main.cuda_stream)
    for k in [k for k, v in _ctx_alias.items() if v == mkey]:
        _ctx_alias.pop(k, None)


def end_of_forward():
    """a forward pass that no backward pass will follow (torch.no_grad: inference, greedy decoding) is over once its side streams are
    joined: drop their aliases now -- with autograd on they live until the train step's join_side_stream (backward nodes of the side
    chain look the pass's context up through them)"""
    if not torch.is_grad_enabled():
        release_side_streams()


def allow_encoder_streams(ok: bool):
    """two compute streams for the pass(es) of the CURRENT stream's context (per context, not per process: one train step switching
    the fork off must not switch it off for another model stepping on another stream)"""
    context().allow_streams = bool(ok)


def encoder_streams_in_use() -> int:
    return ENC_STREAMS if context().allow_streams else 1


_rng_state = {}
_site_counter = [0]


def rng_tensor(device=None) -> torch.Tensor:
    """Per-device {seed, step} pair read by every dropout site (device memory => graph-replayable)."""
    dev = None if device is None else torch.device(device)
    if dev is None or dev.type != "cuda":
        # the dropout stream lives in GPU memory: a host device (a CPU-resident model on its way into a checkpoint) names the current GPU's
        # stream -- never a host tensor cached under a GPU's key, which every later dropout kernel would have been handed
        dev = torch.device("cuda", torch.cuda.current_device())
    elif dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    key = dev.index
    if key not in _rng_state:
        _rng_state[key] = torch.tensor([0x5EED, 0], dtype=torch.int64, device=dev)
    return _rng_state[key]


_rng_seeded = set()       # devices whose dropout stream was seeded explicitly (manual_seed)


def manual_seed(seed: int, device=None):
    t = rng_tensor(device)
    t.copy_(torch.tensor([seed, 0], dtype=torch.int64))
    _rng_seeded.add(str(t.device))


def rng_is_seeded(device=None) -> bool:
    return str(rng_tensor(device).device) in _rng_seeded


def rng_advance():
    """a new forward pass draws new masks"""
    _lib.check(lib.bmt_rng_advance(_p(rng_tensor()), _st()), "bmt_rng_advance")


def add_(out: torch.Tensor, a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """out = a + b, fp32, in one library launch (bmt_add)"""
    _lib.check(lib.bmt_add(_p(a), _p(b), _p(out), out.numel(), _st()), "bmt_add")
    return out


def new_site() -> int:
    """A fresh dropout call-site id (one per module instance and dropout position)."""
    _site_counter[0] += 1
    return _site_counter[0]


# ----------------------------------------------------------------------------- raw launches
def zero_(t: torch.Tensor) -> torch.Tensor:
    """t[...] = 0 in one library launch (bmt_zero): contiguous CUDA tensors whose storage starts on a 16-byte boundary"""
    if not t.is_cuda or not t.is_contiguous() or t.data_ptr() % 16:
        return t.zero_()
    _lib.check(lib.bmt_zero(_p(t), t.numel() * t.element_size(), _st()), "bmt_zero")
    return t


_SPLITK_TARGET = 512     # workgroups a split launch aims for


def _splitk_for(out_rows: int, out_cols: int, red: int) -> int:
    """Split the reduction of a weight-gradient GEMM so the launch fills the 256 CUs (2 workgroups each)."""
    tiles = ((out_rows + 127) // 128) * ((out_cols + 127) // 128)
    want = max(1, _SPLITK_TARGET // tiles)
    return max(1, min(want, (red + 255) // 256))


def _pad64(n: int) -> int:
    return (n + 63) // 64 * 64


# ----------------------------------------------------------------------------- packed rows
# A ragged batch is padded to (B, S, .) and the padded positions carry no information: forward they are only ever read as MASKED keys /
# values (model/multihead_attention.py:17), backward their gradient is exactly zero in every tensor.  Under PACK_ROWS the encoder's streams
# hold the valid rows only, compacted in (b, t) order (bmt_pack_rows builds the layout from the mask -- a hole inside a sequence is as
# good as a padded tail): tensors keep their (B, S, D) SHAPE as a capacity, the number of rows that exist lives in device memory
# (RowPack.rows_ptr) and every row-wise kernel takes min(capacity, that count) when it runs -- the launch sequence stays shape-static
# (hipGraph) over data-dependent extents -- the attention kernels take a sample's rows from RowPack.off[b] and nothing is masked.
# The layout travels with the data: fp32 tensors carry it as ``_bmt_pack``, operand planes as ``Planes.pack``.
PACK_ROWS = _os.environ.get("BMT_PACK_ROWS", "1") != "0"      # A/B switch: "0" = every padded position is computed as the reference does


class RowPack:
    """layout of the valid rows of one modality's batch: ``off`` int32 [2 B + 2] (off[b] = first packed row of sample b, off[B] = number of
    valid rows; the rest is scratch), ``row_map`` int32 [B S] (packed row -> b S + t), ``order`` int32 [B] or None (the samples in the
    length-balanced order the attention kernels walk them: bmt_pack_rows_ordered)"""
    __slots__ = ("off", "row_map", "B", "S", "order")

    def __init__(self, off, row_map, B, S, order=None):
        self.off, self.row_map, self.B, self.S, self.order = off, row_map, B, S, order

    @property
    def cap(self) -> int:
        return self.B * self.S

    @property
    def rows_ptr(self):
        return C.c_void_p(self.off.data_ptr() + 4 * self.B)

    @property
    def off_ptr(self):
        return C.c_void_p(self.off.data_ptr())

    def record_stream(self, stream):
        self.off.record_stream(stream)
        self.row_map.record_stream(stream)
        if self.order is not None:
            self.order.record_stream(stream)


def pack_rows(mask: torch.Tensor) -> RowPack:
    """RowPack of a key-padding mask (B, 1, S) / (B, S), bool or uint8 (bmt_pack_rows; two tiny launches)"""
    m = mask.reshape(mask.shape[0], mask.shape[-1])
    if m.dtype == torch.bool:
        m = m.view(torch.uint8)
    if m.stride(1) != 1:
        m = m.contiguous()
    B, S = m.shape
    off = torch.empty(2 * B + 2, device=m.device, dtype=torch.int32)
    row_map = torch.empty(B * S, device=m.device, dtype=torch.int32)
    order = torch.empty(B, device=m.device, dtype=torch.int32) if (BALANCED_ORDER and 8 < B <= 256) else None
    _lib.check(lib.bmt_pack_rows_ordered(_p(m), m.stride(0), B, S, _p(off), _p(row_map), _p(order), _st()), "bmt_pack_rows_ordered")
    return RowPack(off, row_map, B, S, order)


BALANCED_ORDER = True     # the attention kernels walk the samples of a packed batch in the length-balanced order of its query-side row pack


def _sample_order(qpack, kpack):
    """bmt_attn_*_args.b_order for an attention over packed rows: the query side's order (a sample's cost is (its queries) x (its keys); the
    two modalities' lengths of a video go together -- and any permutation is correct)"""
    pk = qpack if qpack is not None else kpack
    if pk is None or pk.order is None or not BALANCED_ORDER:
        return None
    return C.c_void_p(pk.order.data_ptr())


def pack_of(t) -> Optional["RowPack"]:
    return getattr(t, "_bmt_pack", None)


def carry_pack(pack, *tensors):
    """mark tensors (capacity-shaped) as holding packed rows in ``pack``'s layout"""
    if pack is not None:
        for t in tensors:
            if isinstance(t, torch.Tensor):
                t._bmt_pack = pack
    return tensors[0] if len(tensors) == 1 else tensors


def _rows_dev(pack):
    return pack.rows_ptr if pack is not None else None


def _check_pack(pack, rows: int):
    if pack is not None and pack.cap != rows:
        raise RuntimeError(f"packed rows: a tensor of {rows} rows carries the layout of a batch of {pack.cap}")
    return pack


class Planes:
    """16-bit operand planes of an fp32 [rows, cols] tensor, row stride padded to a multiple of 64 with zeros (the reduction
    extent of the consuming GEMM); any subset of
        hi = bf16(x)          every backward product, and the first plane of a split-bf16 forward operand
        lo = bf16(x - hi)     split-bf16 forward operand (PREC_BF16X3)
        fh = fp16(x)          fp16 forward operand (PREC_F16 / PREC_F16W2)
        fl = fp16(x - fh)     weights of a PREC_F16W2 product"""
    __slots__ = ("hi", "lo", "fh", "fl", "rows", "cols", "pack")

    def __init__(self, hi, lo, rows, cols, fh=None, fl=None, pack=None):
        self.hi, self.lo, self.fh, self.fl, self.rows, self.cols = hi, lo, fh, fl, rows, cols
        self.pack = pack        # RowPack: ``rows`` is the capacity, the rows that exist are counted in device memory (packed rows)

    @property
    def any(self):
        return self.hi if self.hi is not None else self.fh

    def has(self, fmt: str) -> bool:
        return all(getattr(self, n) is not None for n in _FMT[fmt])

    def only(self, *names):
        """a view holding just the named planes (what a backward pass keeps alive)"""
        return Planes(*(getattr(self, n) if n in names else None for n in ("hi", "lo")), self.rows, self.cols,
                      *(getattr(self, n) if n in names else None for n in ("fh", "fl")), pack=self.pack)


# plane sets by consumer: "bwd" single-pass bf16; "x3" split-bf16 activation / weight; "f16" fp16 activation (+ bf16 for the
# backward); "w2" weight of a PREC_F16W2 product (+ bf16 for the k-major dX GEMM)
_FMT = {"bwd": ("hi",), "x3": ("hi", "lo"), "f16": ("hi", "fh"), "w2": ("hi", "fh", "fl"), "f16only": ("fh",)}


def act_fmt(prec: int) -> str:
    """plane set an ACTIVATION needs to be the A operand of a forward product of this precision (and of the backward's dW)"""
    return {PREC_BF16: "bwd", PREC_BF16X3: "x3", PREC_F16: "f16", PREC_F16W2: "f16"}[prec]


def weight_fmt(prec: int) -> str:
    return {PREC_BF16: "bwd", PREC_BF16X3: "x3", PREC_F16: "f16", PREC_F16W2: "w2"}[prec]


def _alloc_planes(rows: int, cols: int, fmt: str, device, ld: Optional[int] = None, zero_pad: bool = False) -> Planes:
    ld = _pad64(cols) if ld is None else ld
    mk = torch.zeros if (zero_pad and ld != cols) else torch.empty
    bufs = {n: mk(rows, ld, device=device, dtype=torch.bfloat16 if n in ("hi", "lo") else torch.float16) for n in _FMT[fmt]}
    return Planes(bufs.get("hi"), bufs.get("lo"), rows, cols, bufs.get("fh"), bufs.get("fl"))


def make_planes(x2: torch.Tensor, fmt: str = "x3", colsum=None, drop=None, gate=None, pack=None) -> Planes:
    """one pass over fp32 x2 [R,C] -> Planes [R][pad64(C)] of the given set; colsum (optional fp32 [C]) += column sums of x2
    (atomic).  drop = (p, site): the planes (and column sums) are those of dropout(x2) with the mask of that site over a
    contiguous [R][C] tensor.  gate = (y [R,C] fp32, scale): those of (y != 0) ? x2 * scale : 0 -- the gradient through a relu (and a
    dropout in front of it) from the saved forward output, without materialising it (bmt_planes_gate)."""
    R, Cc = x2.shape
    pl = _alloc_planes(R, Cc, fmt, x2.device)
    pl.pack = _check_pack(pack, R)
    rd = _rows_dev(pack)
    ld = pl.any.stride(0)
    if gate is not None:
        if pack is not None:
            raise RuntimeError("make_planes: the gate form does not take packed rows")
        y2, gscale = gate
        _lib.check(lib.bmt_planes_gate(_p(x2), x2.stride(0), R, Cc, _p(pl.hi), _p(pl.lo), _p(pl.fh), _p(pl.fl), ld, _p(colsum), _p(y2), y2.stride(0),
                                       float(gscale), _st()), "bmt_planes_gate")
        return pl
    if drop is not None and drop[0] > 0.0:
        _lib.check(lib.bmt_planes_dropout(_p(x2), x2.stride(0), R, Cc, _p(pl.hi), _p(pl.lo), _p(pl.fh), _p(pl.fl), ld, None, None, 0, _p(colsum),
                                          drop[0], _p(rng_tensor()), drop[1], rd, _st()), "bmt_planes_dropout")
    else:
        _lib.check(lib.bmt_planes(_p(x2), x2.stride(0), R, Cc, _p(pl.hi), _p(pl.lo), _p(pl.fh), _p(pl.fl), ld, None, None, 0, _p(colsum), rd, _st()),
                   "bmt_planes")
    return pl


def pad_planes(x3: torch.Tensor, halo: int, tail: int, fmt: str) -> Planes:
    """x fp32 (B,S,C) -> halo-padded planes [B*(S+2*halo) + tail][pad64(C)] (bmt_pad_planes) of the set "bwd" / "x3" / "f16": the
    activation operand of the implicit Conv1d GEMMs.  rows / cols of the returned Planes describe the whole padded buffer."""
    B, S, Cc = x3.shape
    rows = B * (S + 2 * halo) + tail
    pl = _alloc_planes(rows, Cc, fmt, x3.device)
    second = pl.lo if pl.lo is not None else pl.fh
    _lib.check(lib.bmt_pad_planes(_p(x3), B, S, Cc, halo, tail, _p(pl.hi), _p(second), int(pl.fh is not None), pl.hi.stride(0), _st()),
               "bmt_pad_planes")
    return pl


class _WeightPlanes:
    """operand planes of every weight that takes part in a GEMM ([N][pad64(K)], the plane set its site's precision needs), in
    persistent buffers, ALL refreshed by one multi-tensor launch the first time a weight is needed after the optimizer moved
    them (WEIGHT_EPOCH) -- instead of ~200 small launches.  The backward GEMMs read the bf16 hi plane k-major, so no transposed
    copy of a weight exists -- except for the weights of SMALL dX products (a decoder layer's own: get_t), whose row-major form runs
    on the 32 x 32 tile kernel."""

    def __init__(self):
        self.entries = []          # [weakref(owner), key, Planes, fmt, version, detached W]
        self.index = {}            # (id(owner), key) -> position
        self.table = None          # device descriptor table
        self.prefix, self.total_tiles = None, 0      # prefix sums of the tensors' tile counts (the flat launch)
        self._retired = []         # tables a captured hipGraph may still read (its refresh launch has the address baked in): never freed
        self.generation = 0        # bumped when an entry a captured graph may use is replaced or dropped (its planes can be freed then)
        self.fresh_epoch = -1
        self.dirty_table = True
        self.groups = {}           # ids -> [weakrefs, Planes [sum N][Kpad], bias, epoch, fmt, (unused), weakrefs of the biases]
        self.lock = __import__("threading").RLock()      # the registry is shared by every stream / thread of the process

    @staticmethod
    def _key(W):
        return (W.data_ptr(), tuple(W.shape), tuple(W.stride()))

    def _put(self, W, pl, fmt):
        import weakref
        owner = W._base if W._base is not None else W
        entry = [weakref.ref(owner), self._key(W), pl, fmt, None, W.detach(), None, None]      # [6], [7]: transposed bf16 planes hi, lo [K][pad64(N)] (get_t), or None
        pos = self.index.get((id(owner), self._key(W)))
        if pos is not None and pos < len(self.entries) and self.entries[pos][0]() is owner:
            self.entries[pos] = entry          # re-registered (a new plane set, or now as a member of a group): same slot
            self.generation += 1               # the old planes may be freed: graphs captured over them must not be replayed
        else:
            self.entries.append(entry)
            self.index[(id(owner), self._key(W))] = len(self.entries) - 1
        self.dirty_table = True
        return entry

    def _entry(self, W):
        owner = W._base if W._base is not None else W
        pos = self.index.get((id(owner), self._key(W)))
        e = self.entries[pos] if pos is not None and pos < len(self.entries) else None
        return e if (e is not None and e[0]() is owner) else None

    def get_group(self, Ws, bs, fmt):
        """weights that multiply the SAME input (Q/K/V of a self-attention, K/V of a cross-attention) as ONE operand: their
        planes live in adjacent row blocks of one buffer ([sum N][Kpad]), refreshed by the same multi-tensor launch, so the
        three projections are one GEMM forward, one dX GEMM and (with adjacent gradients) one dW GEMM.  Returns (Planes,
        concatenated bias) or None if the shapes do not allow it."""
        import weakref
        K = Ws[0].shape[1]
        if any(W.shape[1] != K or W.shape[0] % 64 != 0 or W.dim() != 2 or not W.is_contiguous() for W in Ws):
            return None
        key = tuple(id(W) for W in Ws)
        g = self.groups.get(key)
        if g is None or any(r() is not W for r, W in zip(g[0], Ws)) or not g[1].has(fmt):
            Nt = sum(W.shape[0] for W in Ws)
            if g is not None and g[1].any.shape[0] == Nt and all(r() is W for r, W in zip(g[0], Ws)):
                fmt = _merge_fmt(fmt, g[4])
            big = _alloc_planes(Nt, K, fmt, Ws[0].device)
            off = 0
            for W in Ws:
                N = W.shape[0]
                sl = lambda t: None if t is None else t[off:off + N]
                self._put(W, Planes(sl(big.hi), sl(big.lo), N, K, sl(big.fh), sl(big.fl)), fmt)
                off += N
            bias = torch.empty(Nt, device=Ws[0].device, dtype=torch.float32) if all(b is not None for b in bs) else None
            g = [[weakref.ref(W) for W in Ws], big, bias, -1, fmt, None, [weakref.ref(b) for b in bs] if bias is not None else None]
            self.groups[key] = g
            if len(self.groups) > 4096:     # models come and go in tests
                self.groups = {k: v for k, v in self.groups.items() if all(r() is not None for r in v[0])}
        stale = self.fresh_epoch != WEIGHT_EPOCH[0] or self.dirty_table
        if not stale:
            for W in Ws:
                stale = stale or self._entry(W)[4] != W._version
        if stale:
            self._refresh_all()
        if g[2] is not None and g[3] != WEIGHT_EPOCH[0] and all(b is not None for b in bs):
            torch.cat([b.detach() for b in bs], out=g[2])
            g[3] = WEIGHT_EPOCH[0]
        return g[1], g[2]

    def _prune(self):
        alive = [e for e in self.entries if e[0]() is not None]
        if len(alive) != len(self.entries):
            self.entries = alive
            self.index = {(id(e[0]()), e[1]): i for i, e in enumerate(alive)}
            self.dirty_table = True
            self.generation += 1               # a retired table still describes the dead weights' (freed) planes

    def _refresh_all(self):
        self._prune()
        if not self.entries:
            return
        if self.dirty_table:
            nb = lib.bmt_planes_desc_bytes()
            host = torch.zeros(len(self.entries), nb, dtype=torch.uint8)
            for i, e in enumerate(self.entries):
                Wd, pl = e[5], e[2]
                _lib.check(lib.bmt_planes_desc(C.c_void_p(host[i].data_ptr()), _p(Wd), Wd.stride(0), Wd.shape[0], Wd.shape[1],
                                               _p(pl.hi), _p(pl.lo), _p(pl.fh), _p(pl.fl), pl.any.stride(0),
                                               _p(e[6]), _p(e[7]), e[6].stride(0) if e[6] is not None else 0),
                           "bmt_planes_desc")
            if self.table is not None:
                self._retired.append((self.table, self.prefix))       # (a few KB each; appended entries leave the old table valid for its graph)
            pre = [0]
            for i in range(len(self.entries)):
                pre.append(pre[-1] + lib.bmt_planes_desc_tiles(C.c_void_p(host[i].data_ptr())))
            dev = self.entries[0][5].device
            self.table = host.to(dev)
            self.prefix = torch.tensor(pre, dtype=torch.int32).to(dev)
            self.total_tiles = pre[-1]
            self.dirty_table = False
        _lib.check(lib.bmt_planes_multi_flat(_p(self.table), _p(self.prefix), len(self.entries), self.total_tiles, _st()), "bmt_planes_multi_flat")
        for e in self.entries:
            e[4] = e[5]._version
        self.fresh_epoch = WEIGHT_EPOCH[0]
        self._refresh_group_biases()
        _rank_refresh_all()          # (the rank-form attentions' W' / c are functions of the weights too: same stream, same moment)

    def _refresh_group_biases(self):
        """the concatenated biases of every fused projection group in ONE launch (they were 14 torch.cat launches per optimizer step, one
        per group at its first use); a group registered later still concatenates its own (get_group)"""
        items, touched = [], []
        for g in self.groups.values():
            ws_ = [r() for r in g[0]]
            if g[2] is None or g[3] == WEIGHT_EPOCH[0] or any(w is None for w in ws_) or g[6] is None:
                continue
            bs_ = [r() for r in g[6]]
            if any(b is None or b.dtype != torch.float32 or not b.is_contiguous() for b in bs_):
                continue
            off = 0
            for b in bs_:
                items.append(_lib.CopyItem(src=b.data_ptr(), dst=g[2].data_ptr() + 4 * off, n=b.numel()))
                off += b.numel()
            touched.append(g)
        if items:
            arr = (_lib.CopyItem * len(items))(*items)
            _lib.check(lib.bmt_copy_multi(arr, len(items), _st()), "bmt_copy_multi")
            for g in touched:
                g[3] = WEIGHT_EPOCH[0]

    def ensure_fresh(self):
        """the once-per-optimizer-step refresh of every registered weight's planes, now (on the current stream)"""
        if self.entries and (self.fresh_epoch != WEIGHT_EPOCH[0] or self.dirty_table):
            self._refresh_all()

    def get_t(self, W, lo=False):
        """the transposed bf16 plane(s) [K][pad64(N)] of a weight (the row-major B operand of a small dX = dY . W; with ``lo`` the split pair of a
        three-pass product against W^T), refreshed with the others"""
        self.get(W, "x3" if lo else "bwd")
        e = self._entry(W)
        if e[6] is None or (lo and e[7] is None):
            if e[2].any._base is not None:
                raise RuntimeError("weight planes: transposed planes of a member of a fused projection group: ask the group (get_group_t)")
            N, K = W.shape
            if e[6] is None:
                e[6] = torch.zeros(K, _pad64(N), device=W.device, dtype=torch.bfloat16)
            if lo and e[7] is None:
                e[7] = torch.zeros(K, _pad64(N), device=W.device, dtype=torch.bfloat16)
            self.dirty_table = True
            self._refresh_all()
        return Planes(e[6], e[7] if lo else None, W.shape[1], W.shape[0])

    def get_group_t(self, Ws, lo=False, bs=None, fmt="bwd"):
        """... of a fused projection group: [K][pad64(sum N)], member i in the column block of its rows in the group's planes"""
        got = self.get_group(Ws, tuple(None for _ in Ws) if bs is None else tuple(bs), "x3" if lo else fmt)
        if got is None:
            return None
        g = self.groups[tuple(id(W) for W in Ws)]
        while len(g) < 8:
            g.append(None)
        if g[5] is None or (lo and g[7] is None) or any(self._entry(W)[6] is None or (lo and self._entry(W)[7] is None) for W in Ws):
            K, Nt = Ws[0].shape[1], sum(W.shape[0] for W in Ws)
            if g[5] is None:
                g[5] = torch.zeros(K, _pad64(Nt), device=Ws[0].device, dtype=torch.bfloat16)
            if lo and g[7] is None:
                g[7] = torch.zeros(K, _pad64(Nt), device=Ws[0].device, dtype=torch.bfloat16)
            off = 0
            for W in Ws:
                e = self._entry(W)
                e[6] = g[5][:, off:off + W.shape[0]]
                e[7] = g[7][:, off:off + W.shape[0]] if g[7] is not None else None
                off += W.shape[0]
            self.dirty_table = True
            self._refresh_all()
        return Planes(g[5], g[7] if lo else None, Ws[0].shape[1], sum(W.shape[0] for W in Ws))

    def get(self, W, fmt):
        e = self._entry(W)
        if e is None or not e[2].has(fmt):
            if e is not None and e[2].any._base is not None:
                raise RuntimeError(f"weight planes: a member of a fused projection group (planes {e[3]}) was asked for '{fmt}' on its own; "
                                   "request the group with that format instead")
            want = fmt if e is None else _merge_fmt(fmt, e[3])
            N, K = W.shape
            e = self._put(W, _alloc_planes(N, K, want, W.device), want)
        if self.fresh_epoch != WEIGHT_EPOCH[0] or e[4] != W._version or self.dirty_table:
            self._refresh_all()
        return e[2]


def _merge_fmt(a: str, b: str) -> str:
    """the smallest plane set that serves both (a weight used under two policies keeps every plane either needs)"""
    names = set(_FMT[a]) | set(_FMT[b])
    for f in ("bwd", "x3", "f16", "w2"):
        if names <= set(_FMT[f]):
            return f
    return "all"


_FMT["all"] = ("hi", "lo", "fh", "fl")
_weights = _WeightPlanes()


def weights_generation() -> int:
    """changes when a registered weight's planes were replaced or a dead model's entries were dropped: a hipGraph captured before
    that has freed plane buffers (or a descriptor table that names them) baked in and must be re-captured, not replayed.  Newly
    REGISTERED weights do not count: the table a graph was captured with is kept alive and stays valid for the weights it names."""
    return _weights.generation


def weights_registry_signature():
    """changes with EVERY change of the registry, registrations included (weights_generation does not count those): equal before and after a
    step = the step found every operand plane, group and table it needed already built"""
    return (_weights.generation, len(_weights.entries), len(_weights.groups), bool(_weights.dirty_table))


def weight_planes(W: torch.Tensor, fmt: str = "x3") -> Planes:
    with _weights.lock:
        _await_planes()
        return _weights.get(W, fmt)


FUSE_PROJECTIONS = True      # Q/K/V (self-attention) and K/V (cross-attention) projections as one GEMM each way


SMALL_DX_OUTPUTS = int(lib.bmt_gemm_small_outputs())      # dX products of at most this many outputs run row-major on the 32 x 32 tile kernel (csrc/gemm_bf16.hip, pipe 5)


def weight_planes_t(W: torch.Tensor) -> Planes:
    with _weights.lock:
        _await_planes()
        return _weights.get_t(W)


def weight_group_t(Ws, lo=False, bs=None):
    if not FUSE_PROJECTIONS:
        return None
    with _weights.lock:
        _await_planes()
        return _weights.get_group_t(tuple(Ws), lo=lo, bs=bs)


def weight_group(Ws, bs, fmt: str = "x3"):
    if not FUSE_PROJECTIONS:
        return None
    with _weights.lock:
        _await_planes()
        return _weights.get_group(tuple(Ws), tuple(bs), fmt)


def group_static_grad(Ws):
    """one [sum N, K] view over the static gradient buffers of Ws if GradientReducer laid them out back to back, else None"""
    gs = [static_grad(W) for W in Ws]
    if any(g is None for g in gs):
        return None
    for a, b in zip(gs[:-1], gs[1:]):
        if a.untyped_storage().data_ptr() != b.untyped_storage().data_ptr() or b.storage_offset() != a.storage_offset() + a.numel():
            return None
    return gs[0].as_strided((sum(W.shape[0] for W in Ws), Ws[0].shape[1]), (Ws[0].shape[1], 1), gs[0].storage_offset())


def fused_weight_groups(model):
    """parameter groups whose gradients should sit back to back (bmt_amd.parallel.GradientReducer(groups=...))"""
    out = []
    for m in model.modules():
        if all(hasattr(m, n) for n in ("linear_Q2d", "linear_K2d", "linear_V2d")):
            out.append([m.linear_Q2d.weight, m.linear_K2d.weight, m.linear_V2d.weight])
    return out


def as_planes(x, fmt: str, pack=None) -> Planes:
    """x: an fp32 [R,C] tensor or Planes; converts (one pass) unless the planes the consumer needs are already there.  pack: the RowPack of
    a tensor that holds packed rows (Planes carry their own)"""
    if isinstance(x, Planes):
        if not x.has(fmt):
            raise RuntimeError(f"operand planes {[n for n in ('hi', 'lo', 'fh', 'fl') if getattr(x, n) is not None]} do not serve a '{fmt}' consumer")
        return x
    return make_planes(x, fmt, pack=pack if pack is not None else pack_of(x))


_SPLITK_WS = {}          # (device index, stream) -> fp32 scratch: launches of one stream are ordered, two streams must not share it
SPLITK_WS_BYTES = 128 << 20


def splitk_workspace(device):
    """scratch for the GEMM's two-pass split-K (bmt_gemm_bf16_args.splitk_ws) of the current stream.  A graph capture runs on its own
    stream: its workspace comes out of the capture's memory pool on first use."""
    key = (torch.device(device).index or 0, torch.cuda.current_stream().cuda_stream)
    ws = _SPLITK_WS.get(key)
    if ws is None:
        ws = torch.empty(SPLITK_WS_BYTES // 4, device=device, dtype=torch.float32)
        _SPLITK_WS[key] = ws
    return ws


AUTO_SPLITK = True       # let the library split the reduction of GEMMs that cannot fill the chip


def _operands(A: Planes, B: Planes, prec: int):
    """(A_hi, A_lo, B_hi, B_lo) pointers tensors of a product of this precision"""
    if prec == PREC_BF16:
        return A.hi, None, B.hi, None
    if prec == PREC_BF16X3:
        return A.hi, A.lo, B.hi, B.lo
    if prec == PREC_F16:
        return A.fh, None, B.fh, None
    if prec == PREC_F16W2:
        return A.fh, None, B.fh, B.fl
    raise ValueError(prec)


def gemm_bf16(A: Planes, B: Planes, C_out, *, precision: int, ldc=0, alpha=1.0, bias=None, relu=False, drop_pre=False, drop_post=False,
              drop_p=0.0, site=0, residual=None, ldr=0, gate=None, gate_scale=1.0, accum=False, splitk=None,
              out_planes: Optional[Planes] = None, a_km: bool = False, b_km: bool = False, conv=None, two_pass: bool = True,
              colsum: Optional[torch.Tensor] = None, a_blk=None):
    """C[M,N] = epilogue(A[M,K] . B[N,K]^T) on operand planes (reduction extents must match and be zero padded).
    a_blk = (n, k): a BLOCK product -- output columns [j n, (j + 1) n) read A's columns [j k, (j + 1) k) (bmt_gemm_bf16_args.a_blk_n): row-major
    B [N][k] (every output block against its own rows of B, k = 128), or b_km B [K][n] (every block its own reduction rows, the same columns).
    a_km / b_km: that operand is given K-MAJOR -- its plane has the reduction index as the row ([K rows][M or N columns]), i.e.
    it is the transpose of what the product needs, read through the hardware transpose unit (single-pass bf16 only).
    out_planes: the result as operand planes -- hi = bf16(c) and ONE of lo = bf16(c - hi) / fh = fp16(c)."""
    ah, al, bh, bl = _operands(A, B, precision)
    if ah is None or bh is None or (precision == PREC_BF16X3 and (al is None or bl is None)) or (precision == PREC_F16W2 and bl is None):
        raise RuntimeError(f"gemm_bf16: operand planes missing for {prec_name(precision)}")
    M = A.cols if a_km else A.rows
    N = B.cols if b_km else B.rows
    Ktrue = 0
    if conv is not None and conv["mode"] == 1:       # implicit Conv1d forward / dX: A = halo-padded activation plane (advanced view)
        M, Kpad = conv["M"], bh.shape[1]
    elif conv is not None and conv["mode"] == 2:     # implicit Conv1d dW: dY and the halo-padded activations, both k-major
        Ktrue, N = A.rows, conv["N"]
        Kpad = _pad64(Ktrue)
    elif a_km or b_km:
        Ktrue = A.rows if a_km else B.rows
        Kpad = _pad64(Ktrue)
        assert (A.rows == Ktrue if a_km else ah.shape[1] == Kpad) and (B.rows == Ktrue if b_km else bh.shape[1] == Kpad), \
            (ah.shape, A.rows, bh.shape, B.rows)
        if a_blk is not None:
            N = (Ktrue // a_blk[1]) * a_blk[0]
    elif a_blk is not None:
        Kpad = a_blk[1]
        assert bh.shape[1] == Kpad and ah.shape[1] == (N // a_blk[0]) * Kpad, (ah.shape, bh.shape, a_blk)
    else:
        Kpad = ah.shape[1]
        assert bh.shape[1] == Kpad, (ah.shape, bh.shape)
    flags = 0
    if bias is not None:
        flags |= EPI_BIAS
    if relu:
        flags |= EPI_RELU
    use_drop = drop_p > 0.0 and (drop_pre or drop_post)
    if use_drop and drop_pre:
        flags |= EPI_DROP_PRE
    if use_drop and drop_post:
        flags |= EPI_DROP_POST
    if residual is not None:
        flags |= EPI_RESIDUAL
    if gate is not None:
        flags |= EPI_GATE
    if accum:
        flags |= EPI_ACCUM
    op = out_planes
    if op is not None and ((op.hi is None and (op.fh is None or op.lo is not None)) or (op.lo is not None and op.fh is not None) or op.fl is not None):
        raise RuntimeError("gemm_bf16: output planes are hi + (lo | fh), or fh alone")
    if splitk is None:
        splitk = 0 if AUTO_SPLITK else 1
    a = GemmBf16Args(_p(ah), _p(al), ah.stride(0), _p(bh), _p(bl), bh.stride(0),
                     _p(C_out), ldc if C_out is not None else N, _p(op.hi) if op else None, _p(op.lo) if op else None,
                     op.any.stride(0) if op else 0, M, N, Kpad, alpha, flags, _p(bias), _p(residual), ldr,
                     _p(gate.hi) if gate is not None else None, gate.hi.stride(0) if gate is not None else 0, gate_scale,
                     drop_p if use_drop else 0.0, _p(rng_tensor()) if use_drop else None, site, precision, splitk)
    a.C_f16 = _p(op.fh) if op else None
    a.a_kmajor, a.b_kmajor, a.K = int(a_km), int(b_km), Ktrue
    a.colsum = _p(colsum)        # += column sums of the (plane-only) output: the bias gradient of the Linear below a dX GEMM
    if a_blk is not None:
        a.a_blk_n, a.a_blk_k = int(a_blk[0]), int(a_blk[1])
    if conv is not None:
        a.N = N
        a.conv_mode, a.conv_cin, a.conv_rows = conv["mode"], conv["cin"], conv["rows"]
        a.conv_S, a.conv_halo = conv.get("S", 1), conv.get("halo", 0)
    # packed rows: the activation operand's layout bounds the product's rows (a weight gradient: its reduction) by a device-side count
    pk = A.pack if A.pack is not None else (B.pack if (a_km and b_km) else None)
    if pk is not None:
        if conv is not None:
            raise RuntimeError("gemm_bf16: packed rows with an implicit convolution")
        _check_pack(pk, Ktrue if a_km else M)
        a.rows_dev = pk.rows_ptr.value
        if op is not None and not a_km:
            op.pack = pk
    if splitk != 1 and two_pass:
        ws = splitk_workspace(ah.device)
        a.splitk_ws, a.splitk_ws_bytes = _p(ws), ws.numel() * 4
    _lib.check(lib.bmt_gemm_bf16(C.byref(a), _st()), "bmt_gemm_bf16")


def linear_fwd(x, W: torch.Tensor, b: Optional[torch.Tensor], out: Optional[torch.Tensor] = None, precision=PREC_BF16X3, **epi):
    """y[M,N] = epilogue(x[M,K] @ W[N,K]^T + b);  x: fp32 tensor or Planes."""
    A = as_planes(x, act_fmt(precision))
    Bw = weight_planes(W, weight_fmt(precision))
    if out is None:
        out = torch.empty(A.rows, W.shape[0], device=W.device, dtype=torch.float32)
    gemm_bf16(A, Bw, out, ldc=out.stride(0), bias=b, precision=precision, **epi)
    return out


def linear_fwd_planes(x, W: torch.Tensor, b: Optional[torch.Tensor], precision=PREC_BF16X3, out_fmt: str = "x3", pad: bool = False, **epi) -> Planes:
    """operand planes (hi + lo | fh, whatever the consumer's ``out_fmt`` names) of epilogue(x @ W^T + b), written straight from
    the GEMM epilogue (no fp32 copy in HBM).  pad=False: row stride N (attention operands); pad=True: row stride pad64(N), zero
    padded (GEMM operands)."""
    N = W.shape[0]
    A = as_planes(x, act_fmt(precision))
    op = _alloc_planes(A.rows, N, out_fmt, W.device, ld=_pad64(N) if pad else N)
    gemm_bf16(A, weight_planes(W, weight_fmt(precision)), None, bias=b, out_planes=op, precision=precision, **epi)
    return op


def linear_dx(dy, W: torch.Tensor, out: Optional[torch.Tensor] = None, **epi):
    """dx[M,K] = dy[M,N] @ W[N,K]   (reduction over N);  dy: fp32 tensor or Planes (hi).  The weight's bf16 plane [N][K] is read
    as stored, k-major: its row IS the reduction index."""
    A = as_planes(dy, "bwd")
    if out is None and epi.get("out_planes") is None:
        out = torch.empty(A.rows, W.shape[1], device=W.device, dtype=torch.float32)
    # (row-major dX through transposed weight planes was measured three times on the ENCODER's products -- rounds 2, 3, 4 -- and never won in
    # the step: DESIGN.md section 6.  The decoder's, <= 1.5 M outputs each, are another regime: 24 ... 80 tiles of 128 x 128 with a split
    # reduction and a second kernel against one launch of 32 x 32 tiles, round 5)
    if A.rows * W.shape[1] <= SMALL_DX_OUTPUTS and W.dim() == 2 and W.is_contiguous() and W.shape[0] <= 2048:      # (the generator's dX reduces over the vocabulary: split)
        gemm_bf16(A, weight_planes_t(W), out, ldc=out.stride(0) if out is not None else 0, precision=PREC_BF16, **epi)
    else:
        gemm_bf16(A, weight_planes(W, "bwd"), out, ldc=out.stride(0) if out is not None else 0, precision=PREC_BF16, b_km=True, **epi)
    return out if out is not None else epi["out_planes"]


# ---- deferred weight gradients: one grouped launch for every dW of a backward pass
# Each dW = dY^T . X (reduction over the batch * sequence rows) accumulates into a static gradient buffer and nothing downstream of
# the backward pass reads it before the optimizer, so the products need not run where autograd reaches them: a training step
# queues them (DEFER_DW) and ``flush_dw`` issues ONE launch over all of them.  Alone a 1024 x 1024 weight is 64 tiles -- the single
# launches split their reductions 8 ways and pay an epilogue kernel and the workspace traffic for it; together the step's ~50
# weight gradients are ~3000 tiles and every reduction runs unsplit.
GROUPED_DW = True        # the queued weight-gradient products of a pass as one grouped launch
_dw_ws = {}


def mark_step_start():
    """call at the beginning of a train step (before zero_grad): records the event that work which depends on nothing but the step's
    beginning is forked from -- the refresh of the weights' operand planes (EARLY_REFRESH)."""
    c = context()
    ev = torch.cuda.Event()
    ev.record()
    c.step_start = ev
    c.planes_ready, c.planes_waited = None, set()
    if EARLY_REFRESH:
        _early_refresh(c, ev)


def clear_step_start():
    c = context()
    c.step_start = None
    c.planes_ready, c.planes_waited = None, set()
    c.deferred, c.deferred_join = None, None      # (a pass that raised before its backward must not leave its zero fill to the next one)


EARLY_REFRESH = True      # the once-per-step refresh of the weights' operand planes runs on its own stream from the step's beginning
_refresh_streams = {}     # device index -> that stream (created outside captures: the eager warm-up steps reach mark_step_start first)


def _early_refresh(c, ev):
    """The refresh (planes_multi_flat_kernel: ~110 us, 202 MB of fp32 weights in, ~300 MB of planes out) used to run where the pass first
    forked a side stream -- alone on the chip, with the pass's own launch-bound prologue (row packs, feature preparation, the first
    LayerNorms: ~90 us of 6 us kernels that read no weight) queued behind it (profiles/r06_n_replay_dispatches.csv, t = 31 ... 143 us).
    Issued here on a stream that waits for nothing but the step's beginning it runs BESIDE that prologue; a stream waits for it where it
    first asks for a weight's planes (_await_planes, from every accessor of the registry).  Only with a built table (the first eager step
    registers weights as it meets them and refreshes in line, as before)."""
    w = _weights
    with w.lock:
        if not w.entries or w.dirty_table or w.fresh_epoch == WEIGHT_EPOCH[0]:
            return
        dev = torch.cuda.current_device()
        rs = _refresh_streams.get(dev)
        if rs is None:
            if torch.cuda.is_current_stream_capturing():
                return
            rs = _refresh_streams[dev] = torch.cuda.Stream(device=dev)
        rs.wait_event(ev)
        with torch.cuda.stream(rs):
            w._refresh_all()
        ready = torch.cuda.Event()
        ready.record(rs)
        c.planes_ready = ready


def _await_planes():
    """the current stream behind the early refresh of this pass, once (called with the registry's lock held, before planes are handed out)"""
    c = context()
    ev = c.planes_ready
    if ev is None:
        return
    cur = torch.cuda.current_stream()
    if cur.cuda_stream in c.planes_waited:
        return
    cur.wait_event(ev)
    c.planes_waited.add(cur.cuda_stream)


CONST_TABLES = True       # a CAPTURED grouped launch reads descriptor tables that were written once, when it was captured
_const_tables = {}        # device index -> {"free": [(device table, pinned image)], "stream": copy stream, "owned": {owner token: [pairs]}}
_CONST_TABLE_BYTES = 256 << 10
_CONST_TABLE_PAIRS = 12


def _const_tables_prepare(idx: int, need: int):
    """(eager launches reach this before any capture does -- the warm-up steps of capture()) table buffers a captured grouped launch can
    OWN: device memory from outside every graph's private pool (inside a capture an allocation may be the memory of a temporary the capture
    freed earlier, and a table written before the replay starts would be overwritten by it), a pinned image to fill, a stream to copy on."""
    pool = _const_tables.get(idx)
    if pool is None:
        pool = _const_tables[idx] = {"free": [], "stream": torch.cuda.Stream(device=idx), "owned": {}}
    size = max(int(need), _CONST_TABLE_BYTES)
    pool["free"] = [pr for pr in pool["free"] if pr[0].numel() >= size]
    n_owned = sum(len(v) for v in pool["owned"].values())
    while len(pool["free"]) < _CONST_TABLE_PAIRS and len(pool["free"]) + n_owned < 8 * _CONST_TABLE_PAIRS:
        pool["free"].append((torch.empty(size, dtype=torch.uint8, device=torch.device("cuda", idx)),
                             torch.empty(size, dtype=torch.uint8, pin_memory=True)))


def release_const_tables(owner):
    """the table buffers of the launches captured under scratch_owner(owner) go back to the pool (its graphs are gone)"""
    for pool in _const_tables.values():
        pool["free"].extend(pool["owned"].pop(owner, []))


def finish_capture():
    """call after a capture, before the first replay: the table images of its grouped launches have arrived"""
    for pool in _const_tables.values():
        pool["stream"].synchronize()


def gemm_bf16_grouped(items, store: bool = False):
    """items: [(dY planes [rows][N_out], X planes [rows][K_in], dW fp32 [N_out][K_in] accumulated in place)] -> one launch.
    store: every item has a placed output and a reduction short enough to stay unsplit (<= 96 stages of 64) -- one writer per element: the
    products are STORED, not accumulated (nobody zeroes the outputs first)"""
    n = len(items)
    arr = (GemmBf16Args * n)()
    for a, item in zip(arr, items):
        A, B, Cm = item[:3]
        if len(item) > 3:              # (first output row, rows that exist) in device memory: the product's output is a packed row range
            a.c_row_dev, a.m_dev = item[3]
        rows = A.rows
        assert B.rows == rows, (A.rows, B.rows)
        a.A_hi, a.lda, a.B_hi, a.ldb = A.hi.data_ptr(), A.hi.stride(0), B.hi.data_ptr(), B.hi.stride(0)
        a.C, a.ldc = Cm.data_ptr(), Cm.stride(0)
        a.M, a.N, a.Kpad, a.K = A.cols, B.cols, _pad64(rows), rows
        a.alpha, a.gate_scale, a.flags, a.precision, a.splitk = 1.0, 1.0, (0 if store else EPI_ACCUM), PREC_BF16, 1
        a.a_kmajor, a.b_kmajor = 1, 1
        assert not store or (len(item) > 3 and _pad64(rows) <= 96 * 64), "gemm_bf16_grouped(store=True): placed outputs with unsplit reductions only"
        pk = A.pack if A.pack is not None else B.pack
        if pk is not None:             # packed rows: the reduction runs over the rows that exist (a device-side count)
            a.rows_dev = _check_pack(pk, rows).rows_ptr.value
    dev = items[0][2].device
    need = int(lib.bmt_gemm_bf16_grouped_ws_bytes(n))
    # the descriptor tables live in device memory until the launch has executed; several grouped launches of one backward pass
    # (flush points) are in flight together, so the scratch buffers rotate (allocated in the eager warm-up steps, before a capture)
    def in_line():       # tables written by kernels in front of the product, into a rotating scratch buffer of the stream
        sk = (dev, torch.cuda.current_stream().cuda_stream)
        slot = _dw_ws.setdefault(sk + ("turn",), [0])
        slot[0] = (slot[0] + 1) % 8
        ws = _dw_ws.get(sk + (slot[0],))
        if ws is None or ws.numel() < need:
            ws = torch.empty(max(need, 64 << 10), dtype=torch.uint8, device=dev)
            _dw_ws[sk + (slot[0],)] = ws
        _lib.check(lib.bmt_gemm_bf16_grouped(arr, n, _p(ws), ws.numel(), _st()), "bmt_gemm_bf16_grouped")

    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if not torch.cuda.is_current_stream_capturing():
        if CONST_TABLES:
            _const_tables_prepare(idx, need)
        return in_line()
    # captured: the ~11 table-writer launches (descriptors in their kernel arguments, a few microseconds each) would execute in front of the
    # product at every replay -- ~55 us before the weight-gradient launch, ~45 us before each memory-gradient launch
    # (profiles/r06_n_replay_dispatches.csv; forked onto another stream they still ran where they were captured,
    # profiles/r06_o_replay_dispatches.csv) -- for tables that are the same at every replay: a captured launch's buffers never move.  So the
    # image is built on the host NOW, copied into a table this launch owns on a stream that is NOT capturing, and the graph holds the
    # product alone (finish_capture() waits for the copies).
    pool = _const_tables.get(idx) if CONST_TABLES else None
    pair = None
    if pool is not None and pool["free"] and pool["free"][-1][0].numel() >= need:
        pair = pool["free"].pop()
        pool["owned"].setdefault(_SCRATCH_OWNER[0] if _SCRATCH_OWNER[0] is not None else "captured", []).append(pair)
    if pair is None:
        return in_line()
    dtab, himg = pair
    launch = (C.c_int * 2)()
    _lib.check(lib.bmt_gemm_bf16_grouped_image(arr, n, C.c_void_p(himg.data_ptr()), himg.numel(), launch), "bmt_gemm_bf16_grouped_image")
    _lib.check(lib.bmt_copy_h2d_async(_p(dtab), C.c_void_p(himg.data_ptr()), need, C.c_void_p(pool["stream"].cuda_stream)), "bmt_copy_h2d_async")
    _lib.check(lib.bmt_gemm_bf16_grouped_run(_p(dtab), n, launch, _st()), "bmt_gemm_bf16_grouped_run")


# the gradient arena's zero fill beside the decoder's forward instead of at the head of the step.  Won its A/B before the rank form (7.019 -> 6.99:
# profiles/r06_q_ab_defer_zero.txt); with it the head of the step waits ~250 us for the rank-form preparations on the refresh stream, the fill
# runs beside them for nothing, and the fork / join around the decoder's entry costs more than it hides: 6.410 / 6.407 -> 6.378 / 6.376 and
# 6.623 / 6.632 -> 6.585 / 6.582 ms on two boxes (profiles/r06_y5_head_ab.txt).  The mechanism (defer_beside) stays for callers with another head.
DEFER_ZERO = False


def defer_beside(fn):
    """``fn()`` issues work of this pass that nothing needs before the backward pass starts (the zero fill of the gradient arena: 202 MB,
    60-100 us of HBM writes at the head of a step, in front of everything -- profiles/r06_p_replay_dispatches.csv).  It runs where the
    pass calls run_deferred_beside() -- the decoder's entry: ~0.6 ms of launch-bound small kernels that leave the chip's bandwidth idle --
    on the auxiliary stream, or in line at join_deferred() if the model never got there."""
    context().deferred = fn


def run_deferred_beside():
    c = context()
    fn, c.deferred = c.deferred, None
    if fn is None:
        return
    aux = _aux_stream()
    if aux is None:
        fn()
        return
    aux.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(aux):
        fn()
    c.deferred_join = aux


def join_deferred():
    """before the backward pass: the deferred launch has been issued, and the current stream is behind it"""
    c = context()
    fn, c.deferred = c.deferred, None
    if fn is not None:
        fn()
    aux, c.deferred_join = c.deferred_join, None
    if aux is not None:
        torch.cuda.current_stream().wait_stream(aux)


COLSUM_BESIDE_DW = True
_aux_streams = {}         # device index -> a stream for small launches that run beside a large one of the same pass (created outside captures)


def _aux_stream():
    dev = torch.cuda.current_device()
    s = _aux_streams.get(dev)
    if s is None:
        if torch.cuda.is_current_stream_capturing():
            return None
        s = _aux_streams[dev] = torch.cuda.Stream(device=dev)
    return s


def flush_dw():
    """issue the queued weight-gradient products (call after the backward pass, before anything reads the gradients)"""
    ctx = context()
    items, ctx.pending_dw = ctx.pending_dw, []
    done, ctx.pending_done = ctx.pending_done, []
    ctx.pending_ids.clear()
    cs, ctx.pending_cs = ctx.pending_cs, []
    side, ctx.pending_side = ctx.pending_side, []

    def run_side():           # one small grouped launch over the side products, then their follow-ups
        if side:
            gemm_bf16_grouped([it for it, _ in side])
            for _, fn in side:
                fn()
    # the queued column sums (LayerNorm / bias partials -> their gradients) touch nothing the weight-gradient products touch: beside the grouped
    # launch on the auxiliary stream instead of alone behind it (45 us with one kernel in flight, profiles/r06_n_replay_dispatches.csv); so do
    # the side products and what follows them (the rank-form attentions' chain rule: ~150 us behind the launch, profiles/r06_rank_f_timeline.txt)
    aux = _aux_stream() if ((cs or side) and len(items) > 1 and GROUPED_DW and COLSUM_BESIDE_DW) else None
    if aux is not None:
        cur = torch.cuda.current_stream()
        aux.wait_stream(cur)
        with torch.cuda.stream(aux):
            if cs:                    # (first: the rank form's chain rule reads dc, whose partial sums are among them)
                _colsum_launch(cs)
            run_side()
        cs, side = [], []
    if items:
        if len(items) == 1 or not GROUPED_DW:
            for dyT, xT, into in items:
                gemm_bf16(dyT, xT, into, ldc=into.stride(0), accum=True, splitk=_splitk_for(dyT.cols, xT.cols, dyT.rows), precision=PREC_BF16,
                          a_km=True, b_km=True)
        else:
            gemm_bf16_grouped(items)
    if aux is not None:
        cur.wait_stream(aux)
    if cs:
        _colsum_launch(cs)
    run_side()
    post, ctx.pending_post = ctx.pending_post, []
    for fn in post:           # launches that read what the products above wrote (side by side on streams of their own they finish no earlier: 6.64 / 6.63
        fn()                  # against 6.64 / 6.62 ms, profiles/r06_rank_ab6.txt)
    for p in done:            # their products are on the stream now: the reducer may launch the bucket's all-reduce behind them
        grad_done(p)
    hs, ctx.gen_handles = ctx.gen_handles, []
    for h in hs:              # the generator's parameters were counted twice (GeneratorFn + FusedGenLossFn) and only the fused node ran
        if not h.ran:
            grad_done(h.W)
            grad_done(h.b)


def _colsum_launch(items):
    """items: (partials tensor, column offset, out tensor, rows, ld, D): out[c] += sum_r partials[r * ld + column offset + c]"""
    arr = (_lib.ColsumItem * len(items))()
    for i, (part, off, out, rows, ld, D) in enumerate(items):
        arr[i] = _lib.ColsumItem(part=part.data_ptr() + 4 * off, out=out.data_ptr(), rows=rows, D=D, ld=ld)
    _lib.check(lib.bmt_colsum_multi(arr, len(items), _st()), "bmt_colsum_multi")


def colsum_deferred(items, params=(), queue: bool = False):
    """the second stage of small reductions whose partial sums are already on the stream (LayerNorm dgamma / dbeta, attention bias
    gradients): queued next to the weight-gradient products while a backward pass defers those (StepContext.defer_dw) and issued by
    flush_dw as ONE launch for the whole pass (~90 reductions per train_cap step, each a 4-us kernel behind a kernel boundary when
    launched on its own); otherwise launched at once.  params: the parameters the results belong to -- their grad_done reports wait
    for the flush, exactly as for queued weight gradients.  The queue keeps the partials' tensors alive."""
    ctx = context()
    params = [p for p in params if p is not None]
    if ctx.defer_dw and (params or queue):       # (a result that is handed back to autograd as a tensor must be complete now; queue: the caller
        ctx.pending_cs.extend(items)             # knows its reader runs behind the flush's column-sum launch -- the rank form's dc)
        ctx.pending_ids.update(id(p) for p in params)
    else:
        _colsum_launch(items)


def linear_dw(dy: Planes, x: Planes, into: Optional[torch.Tensor] = None, params=()) -> Optional[torch.Tensor]:
    """dW[N,K] = dy[M,N]^T @ x[M,K]: both bf16 planes as stored, k-major (their rows are the reduction index).
    into: accumulate straight into this (live) gradient buffer.  params: the parameter(s) ``into`` belongs to -- when the product
    is queued for a grouped launch, their grad_done reports are held back until flush_dw has issued it."""
    ctx = context()
    if ctx.defer_dw and into is not None:
        ctx.pending_dw.append((dy, x, into))
        ctx.pending_ids.update(id(p) for p in params)
        return None
    N, K, M = dy.cols, x.cols, dy.rows
    sk = _splitk_for(N, K, M)
    # (two-pass split-K has one writer per element: no zero-fill, no atomics)
    dW = into if into is not None else torch.empty(N, K, device=dy.hi.device, dtype=torch.float32)
    gemm_bf16(dy, x, dW, ldc=dW.stride(0), accum=into is not None, splitk=sk, precision=PREC_BF16, a_km=True, b_km=True)
    return None if into is not None else dW


def param_of(p: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """the PARAMETER behind ``p``: p itself, or -- for a whole-extent contiguous view of a parameter with a static gradient buffer (the
    [N][K] face ``conv.weight[:, :, 0]`` of a kernel-size-1 Conv1d weight [N][K][1]: the proposal heads' second and third layers) -- that
    parameter.  Without this a view has no static buffer: its dW came back as a tensor and autograd's select backward turned it into
    a zero fill + a strided copy + an accumulation add per layer and step (120 framework launches per configs[3] step, round 6)."""
    if p is None or getattr(p, "_bmt_static_grad", False):
        return p
    b = getattr(p, "_base", None)
    if b is not None and getattr(b, "_bmt_static_grad", False) and p.numel() == b.numel() and p.is_contiguous() and b.is_contiguous() \
            and p.data_ptr() == b.data_ptr():
        return b
    return p


def static_grad(p: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """the persistent gradient buffer of a parameter, if bmt_amd.parallel.GradientReducer bound one (``p.grad`` is then a
    view into a flat bucket that is zeroed once per step): weight / bias gradients are accumulated straight into it by
    the GEMM epilogue / column-sum atomics -- no dW temporary, no zero-fill, no ``grad += dW`` pass.  For a whole-extent view of such a
    parameter (param_of): the buffer in the view's shape."""
    q = param_of(p)
    if q is None or not getattr(q, "_bmt_static_grad", False) or q.grad is None or not q.grad.is_contiguous():
        return None
    return q.grad if q is p else q.grad.view(p.shape)


def note_use(*params):
    """forward-side bookkeeping for fused gradient accumulation: count how many times a parameter with a static gradient
    buffer takes part in this step's graph, so that ``grad_done`` reports it to the reducer only after the LAST of its
    backward contributions (a parameter shared by two modules must not have its bucket all-reduced after the first)."""
    for p in params:   # (counts are reset by GradientReducer.zero_grad, so forwards outside a training step are harmless)
        p = param_of(p)
        if p is not None and getattr(p, "_bmt_static_grad", False):
            p._bmt_uses = getattr(p, "_bmt_uses", 0) + 1


def grad_done(p: Optional[torch.Tensor]):
    """one backward contribution to p has been accumulated into its static buffer; when it was the last one, tell the
    reducer that p's gradient is final (replaces autograd's post-accumulate hook for fused accumulation)"""
    p = param_of(p)
    cb = getattr(p, "_bmt_on_grad", None) if p is not None else None
    if cb is None:
        return
    ctx = context()
    if id(p) in ctx.pending_ids:         # its weight-gradient product is still queued (ops.flush_dw reports it)
        ctx.pending_done.append(p)
        return
    left = getattr(p, "_bmt_uses", 1) - 1
    p._bmt_uses = left
    if left <= 0:
        cb(p)


def wgrad(W, b, dyP: Planes, xP: Planes, dy2_for_bias=None, bias_sum=None):
    # (packed rows: dyP.pack bounds the reduction of the product and of the bias column sums)
    """weight and bias gradient of a Linear: returns (dW, db) tensors for autograd, or (None, None) after accumulating into
    the parameters' static buffers.  bias_sum: an already computed column sum (from grad_planes) or None."""
    gW = static_grad(W)
    dW = linear_dw(dyP, xP, into=gW, params=(param_of(W),))
    if gW is not None:
        grad_done(W)
    db = None
    if b is not None:
        gb = static_grad(b)
        if bias_sum is None:
            bias_sum = colsum(dy2_for_bias, pack=dyP.pack)
        if gb is not None:
            # (atomic accumulation behind the queue of this pass's small reductions: parts of a batch in flight on different streams add to
            # the same buffer)
            colsum_deferred([(bias_sum, 0, gb, 1, bias_sum.numel(), bias_sum.numel())], params=(b,))
            grad_done(b)
        else:
            db = bias_sum
    return dW, db


def drop_grad(dy2: torch.Tensor, b, p: float, site: int):
    """gradient through a dropout site on its way into a Linear's backward: (dy2, (p, site)) when the mask can be applied inside
    the operand conversion (grad_planes), else (dropout(dy2), None)"""
    if p <= 0.0:
        return dy2, None
    if b is None or static_grad(b) is not None:
        return dy2, (p, site)
    return dropout_raw(dy2, p, site), None


def lin_bwd(dy2: torch.Tensor, W, b, x_for_dw, need_dx: bool = True, need_dw: bool = True, drop=None, pack=None, **dx_epi):
    """backward of y = x W^T + b given dy2 [M,N]: one pass builds the gradient's operand plane (+ bias column sums),
    then dX = dY.W and dW += dY^T.X.  x_for_dw: the layer input, fp32 or Planes with a bf16 hi plane.  Returns
    (dx or None, dW or None, db or None); dW/db are None when they were accumulated straight into the parameters' static
    gradient buffers."""
    P, bias_done = grad_planes(dy2, b, drop=drop, pack=pack)
    assert drop is None or bias_done or b is None       # (drop_grad guarantees it: the bias sum must see the masked gradient)
    dx = linear_dx(P, W, **dx_epi) if need_dx else None
    dW = db = None
    if need_dw:
        dW, db = wgrad(W, None if bias_done else b, P, bwd_planes(x_for_dw, pack=pack), dy2_for_bias=dy2)
    elif b is not None and not bias_done:
        db = colsum(dy2, pack=pack)
    return dx, dW, db


def grad_planes(dy2: torch.Tensor, bias: Optional[torch.Tensor] = None, drop=None, pack=None):
    """(bf16 plane of an upstream gradient -- the A operand of dX and, k-major, of dW --, bias-gradient handled?) in ONE pass
    over it; when ``bias`` has a static gradient buffer its column sums are accumulated into that buffer by the same pass.
    drop = (p, site): the gradient first goes through that dropout site's mask (backward of ``x + dropout(y)`` w.r.t. y)."""
    gb = static_grad(bias)
    P = make_planes(dy2, "bwd", colsum=gb, drop=drop, pack=pack)
    if gb is not None:
        grad_done(bias)
    return P, gb is not None


LN_EMIT_GRAD_PLANE = _os.environ.get("BMT_LN_EMIT", "1") != "0"      # switch: "0" = every upstream gradient goes through its own conversion pass again


def request_grad_plane(out: torch.Tensor, p: float, site: int):
    """forward side: ``out = x + dropout_site(sublayer(LN x))`` was written by a sublayer's last GEMM.  Whoever differentiates ``out``'s FIRST
    use -- the next ResidualConnection's LayerNorm -- may hand back, next to d out, the bf16 operand plane of dropout_site(d out) and its
    column partials: exactly what this sublayer's backward would otherwise build in a pass of its own (ResidualNormFn.backward,
    bmt_layernorm_bwd_emit).  A note on the tensor; nobody is obliged to honour it."""
    if LN_EMIT_GRAD_PLANE and isinstance(out, torch.Tensor) and out.is_cuda and out.shape[-1] % 4 == 0:
        out._bmt_gp_req = (float(p), int(site))


def grad_planes_from(dout: torch.Tensor, dy2: torch.Tensor, bias: Optional[torch.Tensor], drop, pack=None):
    """grad_planes(dy2, bias, drop) -- or, when ``dout`` arrives with the plane its producer already built for exactly this dropout site
    (ResidualNormFn.backward), that plane and the queued reduction of its column partials into the bias gradient"""
    gp = getattr(dout, "_bmt_gplane", None)
    if gp is not None:
        pl, p, site, ws, nblk = gp
        want = (p, site) if p > 0.0 else None
        D = dy2.shape[1]
        gb = static_grad(bias)
        if (drop == want or (drop is not None and want is not None and tuple(drop) == want)) and pl.rows == dy2.shape[0] and pl.cols == D \
                and (bias is None or gb is not None) and getattr(dout, "_bmt_gplane_version", -1) == dout._version and pl.pack is pack:
            if bias is not None:
                colsum_deferred([(ws, 2 * D, gb, nblk, 3 * D, D)], params=(bias,))
                grad_done(bias)
            return pl, True
    return grad_planes(dy2, bias, drop=drop, pack=pack)


def bwd_planes(x, pack=None) -> Planes:
    """operand of x for the dW product (its bf16 hi plane, k-major): x fp32 tensor or Planes"""
    if isinstance(x, Planes):
        if x.hi is None:
            raise RuntimeError("the backward needs the bf16 plane of a saved activation")
        return x
    return make_planes(x, "bwd", pack=pack)


def colsum(x2: torch.Tensor, pack=None) -> torch.Tensor:
    M, N = x2.shape
    out = torch.empty(N, device=x2.device, dtype=torch.float32)
    _lib.check(lib.bmt_colsum(_p(x2), x2.stride(0), M, N, _p(out), 0, _rows_dev(_check_pack(pack, M)), _st()), "bmt_colsum")
    return out


def _mask_args(mask: Optional[torch.Tensor], B: int, Sq: int, Sk: int):
    """(tensor kept alive, ptr, batch stride, query stride) for a (B,1,Sk) or (B,Sq,Sk) bool/uint8 mask."""
    if mask is None:
        return None, None, 0, 0
    if mask.dim() == 4:      # (B,1,1,Sk) / (B,1,Sq,Sk) as attention() receives it
        mask = mask.squeeze(1)
    if mask.dtype == torch.bool:
        mask = mask.view(torch.uint8)
    elif mask.dtype != torch.uint8:
        mask = (mask != 0).view(torch.uint8)
    if mask.dim() != 3 or mask.shape[0] != B or mask.shape[2] != Sk or mask.shape[1] not in (1, Sq):
        raise RuntimeError(f"attention mask shape {tuple(mask.shape)} does not match (B={B}, 1|Sq={Sq}, Sk={Sk})")
    if mask.stride(2) != 1:
        mask = mask.contiguous()
    qs = 0 if mask.shape[1] == 1 else mask.stride(1)
    return mask, _p(mask), mask.stride(0), qs


def attn_fwd(q, k, v, mask, H, drop_p=0.0, site=0, precision=PREC_BF16X3):
    """fp32 operands (the module-level ``attention()`` surface): q:(B,Sq,D) k,v:(B,Sk,D) contiguous -> o:(B,Sq,D) (post-dropout),
    lse:(B,H,Sq); the kernel converts while staging."""
    B, Sq, D = q.shape
    Sk = k.shape[1]
    dk = D // H
    o = torch.empty_like(q)
    lse = torch.empty(B, H, Sq, device=q.device, dtype=torch.float32)
    keep, mptr, mbs, mqs = _mask_args(mask, B, Sq, Sk)
    use_drop = drop_p > 0.0
    a = AttnFwdArgs(_p(q), _p(k), _p(v), _p(o), _p(lse), q.stride(1), k.stride(1), v.stride(1), o.stride(1),
                    q.stride(0), k.stride(0), v.stride(0), o.stride(0), mptr, mbs, mqs, B, H, Sq, Sk, dk,
                    1.0 / math.sqrt(dk), drop_p if use_drop else 0.0, _p(rng_tensor()) if use_drop else None, site, precision)
    _lib.check(lib.bmt_attn_fwd(C.byref(a), _st()), "bmt_attn_fwd")
    return o, lse


def attn_bwd(q, k, v, o, do, lse, mask, H, drop_p=0.0):
    B, Sq, D = q.shape
    Sk = k.shape[1]
    dk = D // H
    dq, dk_, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    delta = torch.empty(B, H, Sq, device=q.device, dtype=torch.float32)
    keep, mptr, mbs, mqs = _mask_args(mask, B, Sq, Sk)
    a = AttnBwdArgs(_p(q), _p(k), _p(v), _p(o), _p(do), _p(lse), _p(dq), _p(dk_), _p(dv), _p(delta),
                    q.stride(1), k.stride(1), v.stride(1), o.stride(1), q.stride(0), k.stride(0), v.stride(0), o.stride(0),
                    mptr, mbs, mqs, B, H, Sq, Sk, dk, 1.0 / math.sqrt(dk), drop_p)
    _lib.check(lib.bmt_attn_bwd(C.byref(a), _st()), "bmt_attn_bwd")
    return dq, dk_, dv


def attn_fwd_bf16(qh, ql, kh, kl, vh, vl, mask, H, drop_p=0.0, site=0, precision=PREC_BF16X3):
    """raw kernel call (kernel tests): planes (B,S,D) -- bf16 hi [+ lo], or fp16 for PREC_F16 -- -> o (B,Sq,D) fp32 post-dropout,
    lse (B,H,Sq)."""
    B, Sq, D = qh.shape
    Sk = kh.shape[1]
    dk = D // H
    o = torch.empty(B, Sq, D, device=qh.device, dtype=torch.float32)
    lse = torch.empty(B, H, Sq, device=qh.device, dtype=torch.float32)
    keep, mptr, mbs, mqs = _mask_args(mask, B, Sq, Sk)
    use_drop = drop_p > 0.0
    x3 = precision == PREC_BF16X3
    a = AttnFwdBf16Args(_p(qh), _p(ql) if x3 else None, _p(kh), _p(kl) if x3 else None, _p(vh), _p(vl) if x3 else None,
                        _p(o), _p(lse), qh.stride(1), kh.stride(1), vh.stride(1), o.stride(1),
                        qh.stride(0), kh.stride(0), vh.stride(0), o.stride(0), mptr, mbs, mqs, B, H, Sq, Sk, dk,
                        1.0 / math.sqrt(dk), drop_p if use_drop else 0.0, _p(rng_tensor()) if use_drop else None, site, precision)
    _lib.check(lib.bmt_attn_fwd_bf16(C.byref(a), _st()), "bmt_attn_fwd_bf16")
    return o, lse


ATTN_KMEAN = _os.environ.get("BMT_NO_KMEAN") != "1"      # switch: the dQ correction by (row sum of rounded dS) x mean key


def attn_kmean(kh: torch.Tensor, ldk: int, bsk: int, B: int, Sk: int, D: int, mask_args, f16: bool = False, kpack=None) -> Optional[torch.Tensor]:
    """fp32 [B][D] mean key over the valid keys of a K plane (bmt_attn_kmean; f16: the plane holds fp16), or None when the correction
    is switched off"""
    _, mptr, mbs, mqs = mask_args
    if not ATTN_KMEAN or mqs != 0:       # a mask with a row per query (the decoder's causal self-attention: <= 30 keys, error 0.5 % as it is)
        return None
    out = torch.empty(B, D, device=kh.device, dtype=torch.float32)
    _lib.check(lib.bmt_attn_kmean(_p(kh), ldk, bsk, mptr, mbs, mqs, B, Sk, D, _p(out), int(f16), kpack.off_ptr if kpack is not None else None, _st()),
               "bmt_attn_kmean")
    return out


def attn_bwd_bf16(qh, kh, vh, o, do, lse, mask, H, drop_p=0.0):
    B, Sq, D = qh.shape
    Sk = kh.shape[1]
    dk = D // H
    dq = torch.empty(B, Sq, D, device=qh.device, dtype=torch.float32)
    dk_ = torch.empty(B, Sk, D, device=qh.device, dtype=torch.float32)
    dv = torch.empty(B, Sk, D, device=qh.device, dtype=torch.float32)
    delta = torch.empty(B, H, Sq, device=qh.device, dtype=torch.float32)
    doh = torch.empty(B, Sq, D, device=qh.device, dtype=torch.bfloat16)
    keep, mptr, mbs, mqs = _mask_args(mask, B, Sq, Sk)
    km = attn_kmean(kh, kh.stride(1), kh.stride(0), B, Sk, D, (keep, mptr, mbs, mqs))
    a = AttnBwdBf16Args(_p(qh), _p(kh), _p(vh), _p(o), _p(do), _p(lse), _p(dq), _p(dk_), _p(dv), _p(delta), _p(doh),
                        qh.stride(1), kh.stride(1), vh.stride(1), o.stride(1), qh.stride(0), kh.stride(0), vh.stride(0),
                        o.stride(0), dk_.stride(1), dk_.stride(0), mptr, mbs, mqs, B, H, Sq, Sk, dk, 1.0 / math.sqrt(dk), drop_p)
    a.kmean = _p(km)
    _lib.check(lib.bmt_attn_bwd_bf16(C.byref(a), _st()), "bmt_attn_bwd_bf16")
    return dq, dk_, dv


def _plane_buf(rows: int, cols: int, device, dtype=torch.bfloat16) -> torch.Tensor:
    """16-bit [rows][pad64(cols)] with the pad columns zero (they are reduction padding of the consuming GEMM)"""
    ld = _pad64(cols)
    return (torch.empty if ld == cols else torch.zeros)(rows, ld, device=device, dtype=dtype)


def attn_fwd_planes(q: Planes, k: Planes, v: Planes, B, Sq, Sk, D, mask, H, drop_p=0.0, site=0, precision=PREC_BF16X3, out_fmt: str = "x3",
                    kv_shared: bool = False, scale: Optional[float] = None):
    """attention core over projection planes; the post-dropout output is written as operand planes of the out-projection
    (``out_fmt``: "x3" = bf16 hi + lo, "f16" = bf16 hi + fp16, "bwd" = bf16 hi) -- no fp32 copy.
    precision: PREC_BF16X3 (hi + lo planes of q / k / v), PREC_F16 (their fp16 planes) or PREC_BF16.
    Returns (Planes [B*Sq][pad64(D)], lse)."""
    dk = D // H
    x3, f16 = precision == PREC_BF16X3, precision == PREC_F16
    qa, ka, va = (q.fh, k.fh, v.fh) if f16 else (q.hi, k.hi, v.hi)
    if qa is None or ka is None or va is None or (x3 and (q.lo is None or k.lo is None or v.lo is None)):
        raise RuntimeError(f"attn_fwd_planes: projection planes missing for {prec_name(precision)}")
    dev = qa.device
    oh = _plane_buf(B * Sq, D, dev)
    ol = _plane_buf(B * Sq, D, dev) if out_fmt == "x3" else None
    of = _plane_buf(B * Sq, D, dev, torch.float16) if out_fmt == "f16" else None
    lse = torch.empty(B, H, Sq, device=dev, dtype=torch.float32)
    qpack, kpack = _check_pack(q.pack, B * Sq), _check_pack(k.pack, B * Sk)
    if kpack is not v.pack:
        raise RuntimeError("attn_fwd_planes: keys and values in different row layouts")
    if kpack is not None:
        mask = None          # packed keys: every one of a sample's k_off[b + 1] - k_off[b] rows is a valid key
    keep, mptr, mbs, mqs = _mask_args(mask, B, Sq, Sk)
    use_drop = drop_p > 0.0
    ldq, ldk, ldv, ldop = qa.stride(0), ka.stride(0), va.stride(0), oh.stride(0)
    a = AttnFwdBf16Args(Qh=_p(qa), Ql=_p(q.lo) if x3 else None, Kh=_p(ka), Kl=_p(k.lo) if x3 else None,
                        Vh=_p(va), Vl=_p(v.lo) if x3 else None, O=None, lse=_p(lse),
                        ldq=ldq, ldk=ldk, ldv=ldv, ldo=D, bsq=Sq * ldq, bsk=Sk * ldk, bsv=Sk * ldv, bso=Sq * D,
                        mask=mptr, mask_bs=mbs, mask_qs=mqs, B=B, H=H, Sq=Sq, Sk=Sk, dk=dk, scale=(1.0 / math.sqrt(dk)) if scale is None else float(scale),
                        drop_p=drop_p if use_drop else 0.0, rng=_p(rng_tensor()) if use_drop else None, site=site, precision=precision,
                        Oh=_p(oh), Ol=_p(ol), ldop=ldop, bsop=Sq * ldop, Of=_p(of),
                        q_off=qpack.off_ptr if qpack is not None else None, k_off=kpack.off_ptr if kpack is not None else None,
                        b_order=_sample_order(qpack, kpack), kv_shared=int(kv_shared))
    _lib.check(lib.bmt_attn_fwd_bf16(C.byref(a), _st()), "bmt_attn_fwd_bf16")
    return Planes(oh, ol, B * Sq, D, fh=of, pack=qpack), lse


# switch: "0" keeps the two-kernel backward everywhere; "emit" = the split form that leaves P and dS in HBM workspaces (rounds 3-5);
# default (round 6) = the split form whose key side rebuilds them (attn_bwd_dkvr_kernel: 7 products instead of 5, none of the 2 x Sq x Sk x 2
# bytes per (batch, head) written and read back -- 249 MB moved per encoder attention backward instead of 409, profiles/r06_e_attn_bwd_forms_pmc.txt --
# and none of the 2 x 183 + 52 MB of workspaces per stream).  In the step the two are level: 7.09 vs 7.02 ms when the form was built
# (profiles/r06_d_ab_attn_bwd_forms.txt), 6.96 / 6.97 vs 6.97 / 6.98 ms with the samples walked in balanced order (profiles/r06_aa_ab_attn_bwd_forms.txt)
ATTN_BWD_SPLIT = _os.environ.get("BMT_ATTN_BWD_SPLIT", "recompute") != "0"
ATTN_BWD_RECOMPUTE = _os.environ.get("BMT_ATTN_BWD_SPLIT", "recompute") == "recompute"


_SCRATCH = {}            # (device index, stream handle, capture owner | None, name) -> 1-D tensor: scratch that lives inside ONE library call
_SCRATCH_OWNER = [None]  # the step whose hipGraph capture is running (scratch_owner): a captured launch's scratch belongs to THAT step's graphs


class scratch_owner:
    """``with scratch_owner(token):`` -- scratch requested by captured launches inside belongs to ``token`` (a train step); release_scratch(owner=
    token) frees exactly those buffers.  (Round 5 keyed captured scratch by stream only: one step's uncapture() freed buffers a second
    captured step on the same stream still replayed into.)"""

    def __init__(self, token):
        self.token = token

    def __enter__(self):
        self.prev, _SCRATCH_OWNER[0] = _SCRATCH_OWNER[0], self.token
        return self

    def __exit__(self, *exc):
        _SCRATCH_OWNER[0] = self.prev
        return False


def stream_scratch(name: str, numel: int, dtype, device) -> torch.Tensor:
    """scratch of the CURRENT stream that no kernel reads after the library call that wrote it (the P / dS / scaled-q workspaces of the split
    attention backward: 2 x 183 MB + 52 MB for the audio self-attention): one buffer per (device, stream, name), grown to the largest request
    and re-used by every later call -- launches of a stream are ordered, so the next writer cannot overtake the last reader.  No allocator
    call per launch (an eager step issued 56 of them, each a place for the host to stall between two kernels: round 3's driver-run
    roofline).  A hipGraph capture has its own entries (its stream is capturing: the buffer comes out of the graph's private pool and
    stays with the graph)."""
    dev = torch.device(device)
    cap = torch.cuda.is_current_stream_capturing()
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream,
           (_SCRATCH_OWNER[0] if _SCRATCH_OWNER[0] is not None else "captured") if cap else None, name)
    t = _SCRATCH.get(key)
    if t is None or t.numel() < numel or t.dtype != dtype:
        t = _SCRATCH[key] = torch.empty(max(int(numel), 1), device=dev, dtype=dtype)
    return t[:numel]


def release_scratch(capturing: Optional[bool] = None, owner=None):
    """drop per-stream scratch buffers: all of them, those of captured / eager launches only, or -- ``owner`` -- those captured under
    scratch_owner(owner).  Several GB at configs[1]: two 183-MB P / dS workspaces + 52 MB per stream and launch kind.  Scratch that a hipGraph
    was captured over belongs to that graph: release it only after the graph is gone (a train step's uncapture() releases ITS OWN)."""
    for k in [k for k in _SCRATCH if (owner is not None and k[2] == owner) or (owner is None and (capturing is None or (k[2] is not None) == capturing))]:
        del _SCRATCH[k]


def _attn_split_ws(B, H, Sq, Sk, dk, dev):
    """workspaces of the split attention backward (bmt_attn_bwd_split_ws), or None where the two-kernel form runs (d_k < 128, fewer than
    64 queries: the decoder).  P, dS and the scaled copy of q are scratch of the call (stream_scratch); the per-tile bias partials are read
    by the pass's deferred column-sum launch (colsum_deferred), so they are a plain allocation that lives until then."""
    n = [C.c_int64(0), C.c_int64(0), C.c_int64(0)]
    if lib.bmt_attn_bwd_split_ws(B, H, Sq, Sk, dk, C.byref(n[0]), C.byref(n[1]), C.byref(n[2])) != 0:
        return None
    return (stream_scratch("attn_P", n[0].value, torch.bfloat16, dev), stream_scratch("attn_dS", n[0].value, torch.bfloat16, dev),
            stream_scratch("attn_Qb", n[1].value, torch.bfloat16, dev), torch.empty(n[2].value, device=dev, dtype=torch.float32))


def _attn_rc_ws(B, H, Sq, Sk, dk, dev):
    """workspaces of the RECOMPUTE form of the split attention backward (bmt_attn_bwd_rc_ws): one int of live-query bits and one float of
    max |dO| per (batch, head, 128-query tile) -- scratch of the call -- and the per-tile bias partials (read by the pass's deferred
    column-sum launch: a plain allocation).  None where the form does not apply."""
    n = [C.c_int64(0), C.c_int64(0)]
    if lib.bmt_attn_bwd_rc_ws(B, H, Sq, Sk, dk, C.byref(n[0]), C.byref(n[1])) != 0:
        return None
    return stream_scratch("attn_rc", n[0].value, torch.int32, dev), torch.empty(n[1].value, device=dev, dtype=torch.float32)


def attn_bwd_planes(q: Planes, k: Planes, v: Planes, o: Planes, do, lse, B, Sq, Sk, D, mask, H, drop_p, biases,
                    fuse: Optional[str] = None, kv_shared: bool = False, scale: Optional[float] = None, bias_into=(None, None, None), queue_bias: bool = False):
    """attention backward (single-pass bf16 on the hi planes) with the gradients written as GEMM operands: for each of dq, dk,
    dv the bf16 plane (the A operand of the projection's dX and, k-major, of its dW) and the bias gradient (column sums).
    o: the saved output planes (hi + lo, or hi + fh: delta = rowsum(dO * O) reads the most precise form present).
    biases = (bq, bk, bv): a bias with a static gradient buffer is accumulated in place (returned db is None), otherwise into a
    fresh fp32 [D].  fuse = "qkv" (Sq == Sk) / "kv": the gradients share one plane [M][3D | 2D] (column blocks), the operand of
    the fused projection backward; the combined plane is returned as a 4th element.
    bias_into: per bias, a ZEROED fp32 [D] the caller owns, to take the sums instead of a fresh tensor (the rank form's dc: no fill per call).
    Returns [(P, db)] * 3 (+ [P_all])."""
    dk = D // H
    dev = q.any.device
    Mq, Mk = B * Sq, B * Sk
    qpack, kpack = _check_pack(q.pack, Mq), _check_pack(k.pack, Mk)
    if kpack is not None:
        mask = None          # packed keys: nothing is masked (attn_fwd_planes)
    outs = []
    comb = None
    if fuse in ("qkv", "kv") and D % 64 == 0 and (fuse == "kv" or Mq == Mk):
        n = 3 if fuse == "qkv" else 2
        comb = Planes(_plane_buf(Mk, n * D, dev), None, Mk, n * D, pack=kpack)
    for idx, (M, b) in enumerate(((Mq, biases[0]), (Mk, biases[1]), (Mk, biases[2]))):
        slot = None if comb is None else (idx if fuse == "qkv" else idx - 1)
        hi = comb.hi[:, slot * D:(slot + 1) * D] if (slot is not None and slot >= 0) else _plane_buf(M, D, dev)
        gb = static_grad(b)
        db = None
        if b is not None and gb is None:
            db = bias_into[idx] if bias_into[idx] is not None else torch.zeros(D, device=dev, dtype=torch.float32)
        outs.append((hi, gb if gb is not None else db, db))
    delta = torch.empty(B, H, Sq, device=dev, dtype=torch.float32)
    if isinstance(do, Planes):       # dO already as the bf16 plane the kernels read
        assert do.hi.stride(0) == D and do.rows == Mq, (do.hi.shape, D, Mq)
        doh, do = do.hi, None
    else:
        doh = torch.empty(B, Sq, D, device=dev, dtype=torch.bfloat16)
    keep, mptr, mbs, mqs = _mask_args(mask, B, Sq, Sk)
    f16 = q.hi is None         # q / k / v saved as fp16 planes only (the kernels convert while staging)
    qa, ka, va = (q.fh, k.fh, v.fh) if f16 else (q.hi, k.hi, v.hi)
    ldq, ldk, ldv, ldop = qa.stride(0), ka.stride(0), va.stride(0), o.hi.stride(0)
    (qh_, qb_, _), (kh_, kb_, _), (vh_, vb_, _) = outs
    rc = _attn_rc_ws(B, H, Sq, Sk, dk, dev) if (ATTN_BWD_SPLIT and ATTN_BWD_RECOMPUTE and f16 and mqs == 0) else None
    ws = _attn_split_ws(B, H, Sq, Sk, dk, dev) if (rc is None and ATTN_BWD_SPLIT and f16 and mqs == 0) else None
    # the mean-key correction removes the residue of the bf16-rounded dS (8 significand bits); kept on the split form too, whose dQ runs on
    # fp16 dS with per-query scales (without it one tensor of the deep fixture goes from < 2 % to 4.4 %: DESIGN.md section 2)
    # (kv_shared: ONE key / value plane of width d_k for all heads -- one mean key per sample)
    km = attn_kmean(ka, ldk, Sk * ldk, B, Sk, dk if kv_shared else D, (keep, mptr, mbs, mqs), f16=f16, kpack=kpack)
    a = AttnBwdBf16Args(Qh=_p(qa), Kh=_p(ka), Vh=_p(va), O=None, dO=_p(do), lse=_p(lse), dQ=None, dK=None, dV=None,
                        delta_ws=_p(delta), dOh_ws=_p(doh), ldq=ldq, ldk=ldk, ldv=ldv, ldo=D,
                        bsq=Sq * ldq, bsk=Sk * ldk, bsv=Sk * ldv, bso=Sq * D, dkv_ld=D, dkv_bs=Sk * D,
                        mask=mptr, mask_bs=mbs, mask_qs=mqs, B=B, H=H, Sq=Sq, Sk=Sk, dk=dk, scale=(1.0 / math.sqrt(dk)) if scale is None else float(scale), drop_p=drop_p,
                        Oh=_p(o.hi), Ol=_p(o.lo), ldop=ldop, bsop=Sq * ldop,
                        dQh=_p(qh_), dKh=_p(kh_), dVh=_p(vh_), gq_ld=qh_.stride(0), gq_bs=Sq * qh_.stride(0),
                        gkv_ld=kh_.stride(0), gkv_bs=Sk * kh_.stride(0),
                        dQT=None, dKT=None, dVT=None, gqT_ld=0, gkvT_ld=0,
                        dbq=_p(qb_), dbk=_p(kb_), dbv=_p(vb_), Of=_p(o.fh), kmean=_p(km), qkv_f16=int(f16),
                        q_off=qpack.off_ptr if qpack is not None else None, k_off=kpack.off_ptr if kpack is not None else None,
                        b_order=_sample_order(qpack, kpack), kv_shared=int(kv_shared))
    bias_part = None
    if rc is not None:      # the split backward, recompute form: live bits / max |dO| + per-tile bias partials
        a.rc_ws, a.bias_ws = _p(rc[0]), _p(rc[1])
        bias_part = rc[1]
    elif ws is not None:    # the split backward, emitting form: P / dS / scaled-q workspaces + per-tile bias partials (scratch, freed with this call)
        a.P_ws, a.dS_ws, a.Qb_ws, a.bias_ws = (_p(t) for t in ws)
        bias_part = ws[3]
    elif any(b_ is not None for b_ in (qb_, kb_, vb_)):      # two-kernel form (the decoder's attentions): per-tile bias partials only
        nb = lib.bmt_attn_bwd_bias_ws(B, H, Sq, Sk, dk)
        if nb > 0:
            bias_part = torch.empty(nb, device=dev, dtype=torch.float32)
            a.bias_ws = _p(bias_part)
    if bias_part is not None:        # the partials' sums are issued from here (with the pass's other small reductions when it defers them)
        a.defer_bias = 1
    _lib.check(lib.bmt_attn_bwd_bf16(C.byref(a), _st()), "bmt_attn_bwd_bf16")
    if bias_part is not None:
        rq, rk = B * ((Sq + 127) // 128), B * ((Sk + 127) // 128)
        items = [(bias_part, off * D, buf, rows, D, D) for buf, off, rows in ((qb_, 0, rq), (kb_, rq, rk), (vb_, rq + rk, rk)) if buf is not None]
        static = all(b is None or static_grad(b) is not None for b in biases)
        if items:      # (queue_bias: the sums' only reader runs behind the pass's deferred column-sum launch)
            colsum_deferred(items, params=[b for b in biases if b is not None] if static else (), queue=queue_bias)
    res = []
    for (hi, _, db), M, b, pk in zip(outs, (Mq, Mk, Mk), biases, (qpack, kpack, kpack)):
        if b is not None and db is None:
            grad_done(b)
        res.append((Planes(hi, None, M, D, pack=pk), db))
    if comb is not None:
        res.append(comb)
    return res


def lin_bwd_planes(P: Planes, W, xP: Planes, need_dx: bool = True, **dx_epi):
    """dX = dY.W and dW += dY^T.X from the ready-made bf16 plane of dY (the bias gradient was produced with it)."""
    dx = linear_dx(P, W, **dx_epi) if need_dx else None
    gW = static_grad(W)
    dW = linear_dw(P, xP, into=gW, params=(W,))
    if gW is not None:
        grad_done(W)
    return dx, dW


def dropout_raw(x: torch.Tensor, p: float, site: int) -> torch.Tensor:
    y = torch.empty_like(x)
    _lib.check(lib.bmt_dropout(_p(x), _p(y), x.numel(), p, _p(rng_tensor()), site, _st()), "bmt_dropout")
    return y


# ----------------------------------------------------------------------------- fused residual block
# ResidualConnection computes x + dropout(sublayer(LN(x))).  Fused form (FUSE_RESIDUAL): LN writes the operand planes of its
# output itself (no conversion pass), the sublayer's LAST GEMM adds dropout + residual in its epilogue (no dropout_add pass),
# backward applies the dropout mask while converting the incoming gradient to planes (no dropout pass) and the LayerNorm
# backward adds the residual stream's gradient to its own (no autograd add pass).  The residual is OFFERED to the sublayer
# through a module-level slot; MultiheadedAttention / PositionwiseFeedForward take it, anything else leaves it and the
# ResidualConnection falls back to the separate dropout_add kernel.
LN_PLANES_ONLY = True         # a LayerNorm output asked for as planes (fp32_out=False) is not written as fp32 values as well
FUSE_RESIDUAL = _os.environ.get("BMT_NO_FUSE_RES") != "1"      # switch: "1" = LayerNorm / dropout_add / add as separate kernels


class ResidualOffer:
    __slots__ = ("x", "p", "site", "out", "planes_fmt")

    def __init__(self, x, p, site):
        self.x, self.p, self.site, self.out = x, p, site, None
        self.planes_fmt = None       # the taker also writes its result as operand planes of this format (attached to the result)


def offer_residual(x, p, site, planes_fmt=None) -> ResidualOffer:
    off = ResidualOffer(x, p, site)
    off.planes_fmt = planes_fmt
    context().res_offer = off
    return off


def take_residual() -> Optional[ResidualOffer]:
    """the pending offer, if any (one taker: the slot is cleared)"""
    c = context()
    off, c.res_offer = c.res_offer, None
    return off


def planes_of(t, fmt: str):
    """operand planes attached to an activation by its producer (ResidualNormFn) or by an earlier consumer, if they serve a
    consumer of this format and the tensor has not been written since"""
    pl = getattr(t, "_bmt_planes", None)
    if pl is None or not pl.has(fmt) or getattr(t, "_bmt_planes_version", t._version) != t._version:
        return None
    return pl


def _need_fp32(t):
    """a consumer is about to read the fp32 values of ``t``: not possible for a LayerNorm output that was asked for as planes only"""
    if getattr(t, "_bmt_no_fp32", False):
        raise RuntimeError("this LayerNorm output exists as operand planes only (ResidualConnection(..., fp32_out=False)) and its consumer "
                           "asked for a plane format it was not written in: call the ResidualConnection with fp32_out=True")


def record_stream(t, stream):
    """Tensor.record_stream for ``t`` AND the operand planes attached to it (written by its producer on the producer's stream, read by a
    consumer on ``stream``: the caching allocator must not reuse them for the producer's stream while that reader is pending)"""
    t.record_stream(stream)
    pk = pack_of(t)
    if pk is not None:
        pk.record_stream(stream)
    pl = getattr(t, "_bmt_planes", None)
    if pl is not None:
        for x in (pl.hi, pl.lo, pl.fh, pl.fl):
            if x is not None:
                x.record_stream(stream)


def attach_planes(t, pl: Planes):
    if isinstance(t, torch.Tensor):
        t._bmt_planes, t._bmt_planes_version = pl, t._version


class ResidualNormFn(torch.autograd.Function):
    """x -> (x, LayerNorm(x) [, x again]): the branches of a ResidualConnection leave one node, so that their gradients meet again
    in ONE kernel (dx = g_residual + LN backward(g_norm) [+ g_kv]).  The forward kernel also writes the operand planes of the
    normalised output (bf16 hi + the second plane the sublayer's first GEMM reads: bf16 lo or fp16), which is all that GEMM
    reads.  ``kv_alias``: a third output, x once more, for a consumer outside the ResidualConnection -- the OTHER modality's
    cross-attention reads a stream's post-self-attention value as its key / value input (model/encoders.py:63-79), and autograd
    would add that consumer's gradient to this node's with a kernel of its own (8 adds of 13-34 MB per step at config[1])."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, fmt, fp32_out=True, kv_alias=False):
        note_use(gamma, beta)
        xc = _f32c(x)
        D = xc.shape[-1]
        x2 = xc.view(-1, D)
        rows = x2.shape[0]
        y = torch.empty_like(x2)          # fp32_out False: never written (13-34 MB of stores per encoder LayerNorm nothing would read)
        pack = _check_pack(pack_of(x), rows)
        pl = _alloc_planes(rows, D, fmt, x.device)
        pl.pack = pack
        second = pl.lo if pl.lo is not None else pl.fh
        mean = torch.empty(rows, device=x.device, dtype=torch.float32)
        rstd = torch.empty(rows, device=x.device, dtype=torch.float32)
        _lib.check(lib.bmt_layernorm_fwd_planes(_p(x2), D, _p(gamma), _p(beta), _p(y) if fp32_out else None, D, _p(mean), _p(rstd), _p(pl.hi), _p(second),
                                                int(pl.fh is not None), pl.hi.stride(0), rows, D, eps, _rows_dev(pack), _st()), "bmt_layernorm_fwd_planes")
        ctx.save_for_backward(x2, gamma, mean, rstd)
        ctx.beta = beta
        ctx.pack = pack
        ctx.kv_alias = bool(kv_alias)
        ctx.gp_req = getattr(x, "_bmt_gp_req", None) if D % 4 == 0 else None      # (request_grad_plane: the producer of x wants dropout(dx) as a plane)
        context().last_ln = pl
        if kv_alias:
            return xc.view_as(xc), y.view(xc.shape), xc.view_as(xc)
        return xc.view_as(xc), y.view(xc.shape)

    @staticmethod
    def backward(ctx, g_id, g_n, g_kv=None):
        nout = 7
        x2, gamma, mean, rstd = ctx.saved_tensors
        rows, D = x2.shape
        if g_n is None:
            if g_id is not None and g_kv is not None:
                a, b = _f32c(g_id), _f32c(g_kv)
                out = torch.empty_like(a)
                _lib.check(lib.bmt_add(_p(a), _p(b), _p(out), a.numel(), _st()), "bmt_add")
                return (out,) + (None,) * (nout - 1)
            return (g_id if g_id is not None else g_kv,) + (None,) * (nout - 1)
        dy2 = _f32c(g_n).view(rows, D)
        add = _f32c(g_id).view(rows, D) if g_id is not None else None
        add2 = _f32c(g_kv).view(rows, D) if g_kv is not None else None
        if add is None and add2 is not None:
            add, add2 = add2, None
        dx = torch.empty_like(x2)
        beta = ctx.beta
        sg, sb = static_grad(gamma), static_grad(beta)
        fused = sg is not None and sb is not None
        dg = sg if fused else torch.zeros(D, device=x2.device, dtype=torch.float32)
        db = sb if fused else torch.zeros(D, device=x2.device, dtype=torch.float32)
        nblk = max(1, lib.bmt_layernorm_bwd_blocks(rows))
        req = ctx.gp_req if LN_EMIT_GRAD_PLANE else None
        pack = ctx.pack
        rd = _rows_dev(pack)
        gplane, wld, rc = None, 2 * D, -1
        if req is not None:      # dx AND the operand plane of dropout_site(dx) for the sublayer that produced x (+ its column partials)
            ws = torch.empty(nblk * 3 * D, device=x2.device, dtype=torch.float32)
            gph = torch.empty(rows, _pad64(D), device=x2.device, dtype=torch.bfloat16)      # (the kernel writes the pad columns too)
            use_drop = req[0] > 0.0
            rc = lib.bmt_layernorm_bwd_emit(_p(dy2), D, _p(x2), D, _p(gamma), _p(mean), _p(rstd), _p(dx), D, _p(add), D, _p(add2), D, _p(ws), _p(gph), _pad64(D),
                                            req[0] if use_drop else 0.0, _p(rng_tensor()) if use_drop else None, req[1], rows, D, rd, _st())
            if rc == 0:
                gplane, wld = (Planes(gph, None, rows, D, pack=pack), req[0], req[1], ws, nblk), 3 * D
            elif rc != 1:
                _lib.check(rc, "bmt_layernorm_bwd_emit")
        if rc != 0:
            ws = torch.empty(nblk * 2 * D, device=x2.device, dtype=torch.float32)
            if add2 is not None:
                rc = lib.bmt_layernorm_bwd_partial2(_p(dy2), D, _p(x2), D, _p(gamma), _p(mean), _p(rstd), _p(dx), D, _p(add), D, _p(add2), D, _p(ws), rows, D, rd, _st())
                if rc == 1:            # (shapes the vector kernel does not take: one addend by the kernel, the other by a separate add)
                    tmp = torch.empty_like(add)
                    _lib.check(lib.bmt_add(_p(add), _p(add2), _p(tmp), add.numel(), _st()), "bmt_add")
                    add, add2 = tmp, None
            if add2 is None:
                rc = lib.bmt_layernorm_bwd_partial(_p(dy2), D, _p(x2), D, _p(gamma), _p(mean), _p(rstd), _p(dx), D, _p(add), D, _p(ws), rows, D, rd, _st())
        if rc == 0:        # dgamma / dbeta partials of the kernel's workgroups are in ws: their sums join the pass's other small reductions
            colsum_deferred([(ws, 0, dg, nblk, wld, D), (ws, D, db, nblk, wld, D)], params=(gamma, beta) if fused else ())
        else:
            if rc != 1:
                _lib.check(rc, "bmt_layernorm_bwd_partial")
            _lib.check(lib.bmt_layernorm_bwd_add(_p(dy2), D, _p(x2), D, _p(gamma), _p(mean), _p(rstd), _p(dx), D, _p(add), D, _p(dg), _p(db),
                                                 _p(ws), rows, D, rd, _st()), "bmt_layernorm_bwd_add")
        dx = dx.view(g_n.shape)
        if gplane is not None:
            dx._bmt_gplane, dx._bmt_gplane_version = gplane, dx._version
        if fused:
            grad_done(gamma)
            grad_done(beta)
            return (dx,) + (None,) * (nout - 1)
        return (dx, dg, db) + (None,) * (nout - 3)


def residual_norm(x, gamma, beta, eps, prec: int = PREC_BF16X3, fp32_out: bool = True, kv_alias: bool = False):
    """(x passed through, LayerNorm(x) carrying its operand planes as ``_bmt_planes`` [, x once more for an outside consumer]); prec: the
    forward precision of the sublayer's first GEMM.  fp32_out False: the normalised tensor's fp32 values are not written -- for sublayers
    that read the planes (MultiheadedAttention, PositionwiseFeedForward); a consumer that would need the values raises (_need_fp32).
    kv_alias: see ResidualNormFn -- the third tensor carries the operand planes attached to ``x`` (a self-attention's result written as the
    other chain's key / value operand)."""
    outs = ResidualNormFn.apply(x, gamma, beta, eps, act_fmt(prec), fp32_out, kv_alias)
    carry_pack(pack_of(x), *outs)
    xid, xn = outs[0], outs[1]
    c = context()
    attach_planes(xn, c.last_ln)
    c.last_ln = None
    if not fp32_out:
        xn._bmt_no_fp32 = True
    if kv_alias:
        xkv = outs[2]
        pl = getattr(x, "_bmt_planes", None)
        if pl is not None and getattr(x, "_bmt_planes_version", x._version) == x._version:
            attach_planes(xkv, pl)
        return xid, xn, xkv
    return xid, xn


# ----------------------------------------------------------------------------- autograd functions
class LayerNormFn(torch.autograd.Function):
    """nn.LayerNorm over the last dim (model/blocks.py:127,131)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, fmt=None):
        """fmt: also write the result's operand planes of that format from the same kernel (left in StepContext.last_ln for the caller to attach:
        blocks.layer_norm) -- the bridge's Linear (model/blocks.py:149-153) reads them instead of converting the fp32 result in a pass of its own"""
        note_use(gamma, beta)
        xc = _f32c(x)
        D = xc.shape[-1]
        x2 = xc.view(-1, D)
        rows = x2.shape[0]
        y = torch.empty_like(x2)
        mean = torch.empty(rows, device=x.device, dtype=torch.float32)
        rstd = torch.empty(rows, device=x.device, dtype=torch.float32)
        if fmt is not None and D % 4 == 0:
            pl = _alloc_planes(rows, D, fmt, x.device)
            second = pl.lo if pl.lo is not None else pl.fh
            _lib.check(lib.bmt_layernorm_fwd_planes(_p(x2), D, _p(gamma), _p(beta), _p(y), D, _p(mean), _p(rstd), _p(pl.hi), _p(second),
                                                    int(pl.fh is not None), pl.hi.stride(0), rows, D, eps, None, _st()), "bmt_layernorm_fwd_planes")
            context().last_ln = pl
        else:
            _lib.check(lib.bmt_layernorm_fwd(_p(x2), D, _p(gamma), _p(beta), _p(y), D, _p(mean), _p(rstd), rows, D, eps, _st()),
                       "bmt_layernorm_fwd")
        ctx.save_for_backward(x2, gamma, mean, rstd)
        ctx.beta = beta
        return y.view(xc.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, gamma, mean, rstd = ctx.saved_tensors
        rows, D = x2.shape
        dy2 = _f32c(dy).view(rows, D)
        dx = torch.empty_like(x2)
        beta = ctx.beta
        sg, sb = static_grad(gamma), static_grad(beta)
        fused = sg is not None and sb is not None          # accumulate straight into the static gradient buffers
        dg = sg if fused else torch.zeros(D, device=x2.device, dtype=torch.float32)
        db = sb if fused else torch.zeros(D, device=x2.device, dtype=torch.float32)
        nblk = max(1, lib.bmt_layernorm_bwd_blocks(rows))
        ws = torch.empty(nblk * 2 * D, device=x2.device, dtype=torch.float32)
        rc = lib.bmt_layernorm_bwd_partial(_p(dy2), D, _p(x2), D, _p(gamma), _p(mean), _p(rstd), _p(dx), D, None, 0, _p(ws), rows, D, None, _st())
        if rc == 0:
            colsum_deferred([(ws, 0, dg, nblk, 2 * D, D), (ws, D, db, nblk, 2 * D, D)], params=(gamma, beta) if fused else ())
        else:
            if rc != 1:
                _lib.check(rc, "bmt_layernorm_bwd_partial")
            _lib.check(lib.bmt_layernorm_bwd(_p(dy2), D, _p(x2), D, _p(gamma), _p(mean), _p(rstd), _p(dx), D, 0, _p(dg), _p(db),
                                             _p(ws), rows, D, _st()), "bmt_layernorm_bwd")
        if fused:
            grad_done(gamma)
            grad_done(beta)
            return dx.view(dy.shape), None, None, None, None
        return dx.view(dy.shape), dg, db, None, None


class DropoutAddFn(torch.autograd.Function):
    """x + dropout(sub)   (ResidualConnection.forward model/blocks.py:134-136)."""

    @staticmethod
    def forward(ctx, x, sub, p, site):
        xc, sc = _f32c(x), _f32c(sub)
        out = torch.empty_like(xc)
        _lib.check(lib.bmt_dropout_add(_p(sc), _p(xc), _p(out), xc.numel(), p, _p(rng_tensor()) if p > 0 else None, site, _st()),
                   "bmt_dropout_add")
        ctx.p, ctx.site = p, site
        return out

    @staticmethod
    def backward(ctx, dy):
        dyc = _f32c(dy)
        dsub = dropout_raw(dyc, ctx.p, ctx.site) if ctx.p > 0 else dyc
        return dyc, dsub, None, None


class DropoutFn(torch.autograd.Function):
    """standalone dropout with the library RNG (nn.Dropout sites that are not fused anywhere)."""

    @staticmethod
    def forward(ctx, x, p, site):
        ctx.p, ctx.site = p, site
        return dropout_raw(_f32c(x), p, site) if p > 0 else x

    @staticmethod
    def backward(ctx, dy):
        return (dropout_raw(_f32c(dy), ctx.p, ctx.site) if ctx.p > 0 else dy), None, None


class LinearActFn(torch.autograd.Function):
    """y = act(dropout?(x W^T + b))   act in {none, relu}; dropout before (bridge, blocks.py:151-153) or
    after (FFN hidden, blocks.py:168-171) the ReLU, fused in the GEMM epilogue."""

    @staticmethod
    def forward(ctx, x, W, b, relu, drop_mode, p, site, out_fmt=None):
        """out_fmt (round 6): also write the result's operand planes in the epilogue and attach them to it -- the next Linear of a chain (the
        proposal heads' 1-tap layers) finds them instead of converting the fp32 result in a pass of its own; and the input's bf16 plane,
        not its fp32 values, is what the backward keeps for the weight gradient (no second conversion pass there either)."""
        note_use(W, b)
        xc = _f32c(x)
        K = xc.shape[-1]
        x2 = xc.view(-1, K)
        prec = policy_of(None).gemm
        xp = planes_of(x, act_fmt(prec))
        if xp is None or xp.rows != x2.shape[0] or xp.pack is not None:
            xp = make_planes(x2, act_fmt(prec))
        epi = {}
        opl = None
        if out_fmt is not None and W.shape[0] % 64 == 0:
            opl = _alloc_planes(x2.shape[0], W.shape[0], out_fmt, x2.device)
            epi["out_planes"] = opl
        y = linear_fwd(xp, W, b, precision=prec, relu=relu, drop_pre=(drop_mode == "pre"), drop_post=(drop_mode == "post"),
                       drop_p=p, site=site, **epi)
        ctx.relu, ctx.drop_mode, ctx.p, ctx.site = relu, drop_mode, p, site
        ctx.has_bias = b is not None
        ctx.params = (W, b)
        ctx.xdims = (x2.shape[0], K)
        ctx.save_for_backward(xp.hi, W, y if (relu or (p > 0 and drop_mode != "none")) else None)
        out = y.view(*xc.shape[:-1], W.shape[0])
        if opl is not None:
            attach_planes(out, opl)
        return out

    @staticmethod
    def backward(ctx, dy):
        xh, W, y = ctx.saved_tensors
        x2 = Planes(xh, None, ctx.xdims[0], ctx.xdims[1])
        N = W.shape[0]
        dy2 = _f32c(dy).view(-1, N)
        p = ctx.p if ctx.drop_mode != "none" else 0.0
        if ctx.relu:
            # dz = (y != 0) ? dy / (1 - p) : 0 never exists: its bf16 plane and column sums (the bias gradient) come out of one pass over dy and y
            Wp, bp = ctx.params
            bp = bp if ctx.has_bias else None
            gb = static_grad(bp)
            cs = gb if gb is not None else (torch.zeros(N, device=dy2.device, dtype=torch.float32) if bp is not None else None)
            P = make_planes(dy2, "bwd", colsum=cs, gate=(y, 1.0 / (1.0 - p) if p > 0 else 1.0))
            if gb is not None:
                grad_done(bp)
            dx = linear_dx(P, Wp) if ctx.needs_input_grad[0] else None
            dW = None
            if ctx.needs_input_grad[1]:
                dW, _ = wgrad(Wp, None, P, bwd_planes(x2))
            if dx is not None:
                dx = dx.view(*dy.shape[:-1], W.shape[1])
            return dx, dW, (None if gb is not None else cs), None, None, None, None, None
        if p > 0:
            dz = dropout_raw(dy2, p, ctx.site)
        else:
            dz = dy2
        Wp, bp = ctx.params
        dx, dW, db = lin_bwd(dz, Wp, bp if ctx.has_bias else None, x2, need_dx=ctx.needs_input_grad[0],
                             need_dw=ctx.needs_input_grad[1])
        if dx is not None:
            dx = dx.view(*dy.shape[:-1], W.shape[1])
        return dx, dW, db, None, None, None, None, None


class FFNFn(torch.autograd.Function):
    """fc2(dropout(relu(fc1(x))))   PositionwiseFeedForward.forward model/blocks.py:167-174.
    The hidden activation only ever exists as 16-bit operand planes (written by fc1's epilogue, read by fc2 and by the
    backward); backward applies the relu/dropout derivative inside the dH GEMM epilogue (gate on the saved hidden) and dH itself
    only ever exists as a bf16 plane.  pol: the Policy of the enclosing layer."""

    @staticmethod
    def forward(ctx, x, W1, b1, W2, b2, p, site, pol, res=None, res_p=0.0, res_site=0, out_fmt=None):
        note_use(W1, b1, W2, b2)
        xc = _f32c(x)
        x2 = xc.view(-1, xc.shape[-1])
        prec = pol.ffn1
        fmt = act_fmt(prec)
        pack = _check_pack(pack_of(x), x2.shape[0])
        xp = planes_of(x, fmt)            # LayerNorm wrote the operand planes of its output already
        if xp is None:
            _need_fp32(x)
        if xp is None:
            xp = make_planes(x2, fmt, pack=pack)
        if xp.pack is not pack:
            raise RuntimeError("FFNFn: the operand planes attached to the input are in another row layout than the input")
        h = linear_fwd_planes(xp, W1, b1, precision=prec, out_fmt=fmt, pad=True, relu=True, drop_post=True, drop_p=p, site=site)
        epi = {}
        if res is not None:              # x_res + dropout(fc2(h)) in fc2's epilogue (ResidualConnection)
            r2 = _f32c(res).view(-1, W2.shape[0])
            epi = dict(residual=r2, ldr=r2.stride(0), drop_post=True, drop_p=res_p, site=res_site)
        opl = None
        if out_fmt is not None and W2.shape[0] % 4 == 0:        # the reader of this result (the decoder: its memory; the generator) wants it as planes (padded to 64 columns, zeros)
            opl = _alloc_planes(x2.shape[0], W2.shape[0], out_fmt, xc.device)
            epi["out_planes"] = opl
        y = linear_fwd(h, W2, b2, precision=pol.ffn2, **epi)
        ctx.p = p
        ctx.h = h.only("hi")             # the backward reads bf16 planes only
        ctx.xp = xp.only("hi")
        ctx.pack = pack
        ctx.res = (res is not None, res_p, res_site)
        ctx.params = (W1, b1, W2, b2)
        ctx.save_for_backward(W1, W2)
        y = y.view(*xc.shape[:-1], W2.shape[0])
        carry_pack(pack, y)
        if opl is not None:
            attach_planes(y, opl)
        if res is not None:
            request_grad_plane(y, res_p, res_site)
        return y

    @staticmethod
    def backward(ctx, dy):
        W1, W2 = ctx.saved_tensors
        h = ctx.h
        dy2 = _f32c(dy).view(-1, W2.shape[0])
        gscale = 1.0 / (1.0 - ctx.p) if ctx.p > 0 else 1.0
        W1p, b1p, W2p, b2p = ctx.params
        has_res, res_p, res_site = ctx.res
        drop = None
        if has_res:
            dy2, drop = drop_grad(dy2, b2p, res_p, res_site)
        # dH never exists in fp32: the fc2 dX GEMM writes its bf16 plane (relu / dropout derivative applied to whole row
        # segments from the saved hidden plane) and its column sums -- fc1's bias gradient -- from the same epilogue
        P2, b2_done = grad_planes_from(dy, dy2, b2p, drop, pack=ctx.pack) if has_res else grad_planes(dy2, b2p, drop=drop, pack=ctx.pack)
        M_, Dff = h.rows, h.cols
        gb1 = static_grad(b1p)
        cs = gb1 if gb1 is not None else torch.zeros(Dff, device=dy2.device, dtype=torch.float32)
        dhP = Planes(torch.empty(M_, _pad64(Dff), device=dy2.device, dtype=torch.bfloat16), None, M_, Dff)
        linear_dx(P2, W2p, out_planes=dhP, gate=h, gate_scale=gscale, colsum=cs if b1p is not None else None)
        dW2, db2 = wgrad(W2p, None if b2_done else b2p, P2, h, dy2_for_bias=dy2)
        db1 = None
        if b1p is not None:
            if gb1 is not None:
                grad_done(b1p)
            else:
                db1 = cs
        dx, dW1 = lin_bwd_planes(dhP, W1p, ctx.xp, need_dx=ctx.needs_input_grad[0])
        if dx is not None:
            dx = dx.view(*dy.shape[:-1], W1.shape[1])
        return dx, dW1, db1, dW2, db2, None, None, None, (dy if has_res else None), None, None, None


def project_group(Xp: Planes, Ws, bs, prec: int, out_fmt: str):
    """projections that read the same input as ONE GEMM over adjacent weight planes (q|k|v or k|v column blocks of one plane
    buffer, which the attention kernels address with the buffer's row stride); None if the weights cannot be grouped"""
    grp = weight_group(Ws, bs, weight_fmt(prec))
    if grp is None:
        return None
    gst, gb = grp
    Nt = gst.rows
    big = _alloc_planes(Xp.rows, Nt, out_fmt, Xp.any.device, ld=Nt)
    gemm_bf16(Xp, gst, None, bias=gb, out_planes=big, precision=prec)
    outs, off = [], 0
    for W in Ws:
        N = W.shape[0]
        sl = lambda t: None if t is None else t[:, off:off + N]
        outs.append(Planes(sl(big.hi), sl(big.lo), Xp.rows, N, sl(big.fh), sl(big.fl), pack=Xp.pack))
        off += N
    return outs


def attn_operand_fmt(attn_prec: int) -> str:
    """plane set the projections write for q / k / v: the attention forward's operands + the bf16 planes of its backward"""
    return {PREC_BF16X3: "x3", PREC_F16: "f16", PREC_BF16: "bwd"}[attn_prec]


QKV_F16_ONLY = True      # q / k / v as fp16 planes only where the backward can convert them


def attn_train_fmt(attn_prec: int, dk: int) -> str:
    """q / k / v planes of a TRAINING pass: under the fp16 attention policy with d_k >= 128 the fp16 plane alone -- the backward
    kernels convert it to bf16 while staging (bmt_attn_bwd_bf16_args.qkv_f16), the projections write 2 instead of 4 bytes per element"""
    if attn_prec == PREC_F16 and dk >= 128 and QKV_F16_ONLY:
        return "f16only"
    return attn_operand_fmt(attn_prec)


def mha_infer(Q, K, mask, Wq, bq, Wk, bk, Wv, bv, Wo, bo, H, cache, key, pol):
    """MultiheadedAttention.forward for inference with K is V (cross-attention over the encoder memory): the key / value
    projections are taken from ``cache`` when they were computed for the same memory tensor before (greedy decoding re-uses
    them for every generated token; the reference re-encodes the video and re-projects the memory per token,
    epoch_loops/captioning_epoch_loops.py:58-61)."""
    Qc = _f32c(Q)
    B, Sq, Dq = Qc.shape
    D = Wq.shape[0]
    Sk = K.shape[1]
    qkv_fmt = attn_operand_fmt(pol.attn)
    ent = cache.get(key)
    if ent is None or ent[0] is not K:
        Kc = _f32c(K)
        Kp = make_planes(Kc.view(-1, Kc.shape[-1]), act_fmt(pol.kv_gemm))
        r = project_group(Kp, (Wk, Wv), (bk, bv), pol.kv_gemm, qkv_fmt)
        if r is None:
            r = (linear_fwd_planes(Kp, Wk, bk, precision=pol.kv_gemm, out_fmt=qkv_fmt),
                 linear_fwd_planes(Kp, Wv, bv, precision=pol.kv_gemm, out_fmt=qkv_fmt))
        ent = (K, r[0], r[1])
        cache[key] = ent
    k, v = ent[1], ent[2]
    Qp = planes_of(Q, act_fmt(pol.gemm))
    if Qp is None:
        _need_fp32(Q)
        Qp = make_planes(Qc.view(-1, Dq), act_fmt(pol.gemm))
    q = linear_fwd_planes(Qp, Wq, bq, precision=pol.gemm, out_fmt=qkv_fmt)
    o, _ = attn_fwd_planes(q, k, v, B, Sq, Sk, D, mask, H, precision=pol.attn, out_fmt=act_fmt(pol.gemm))
    return linear_fwd(o, Wo, bo, precision=pol.gemm).view(B, Sq, Dq)


class MHAFn(torch.autograd.Function):
    """MultiheadedAttention.forward model/multihead_attention.py:55-86: three input projections, the masked
    softmax-attention core with dropout on its OUTPUT (:22-23), head merge and output projection.

    Every tensor between the GEMMs and the attention kernels exists only as 16-bit operand planes: the projections write
    q/k/v planes from their epilogue (the attention forward's operand format + the bf16 plane its backward reads), the attention
    forward writes the planes of its output, the attention backward writes dq/dk/dv as (bf16 plane, bias sums).  The only fp32
    intermediates are the module's input/output.  pol: the Policy of the enclosing layer (ops.POLICIES)."""

    @staticmethod
    def forward(ctx, Q, K, V, mask, Wq, bq, Wk, bk, Wv, bv, Wo, bo, H, p, site, pol, res=None, res_p=0.0, res_site=0, out_fmt=None):
        note_use(Wq, bq, Wk, bk, Wv, bv, Wo, bo)
        Qc, Kc, Vc = _f32c(Q), _f32c(K), _f32c(V)
        B, Sq, Dq = Qc.shape
        Sk = Kc.shape[1]
        D = Wq.shape[0]
        same_qk, same_kv = Q is K, K is V
        prec_q = pol.gemm
        prec_kv = pol.gemm if same_qk else pol.kv_gemm       # cross-attention: K / V project the (long) other stream
        qkv_fmt = attn_train_fmt(pol.attn, D // H)

        # each distinct input: operand planes of the format its projection reads, holding the bf16 plane the dW product needs
        # -- one pass (none at all when the producer -- LayerNorm -- attached the planes of its output)
        qpack, kpack = _check_pack(pack_of(Q), B * Sq), _check_pack(pack_of(K), B * Sk)
        if pack_of(V) is not kpack:
            raise RuntimeError("MHAFn: keys and values in different row layouts")

        def split(orig, x3d, prec):
            pl = planes_of(orig, act_fmt(prec))
            if pl is None:       # kept on the tensor: the encoder memory is the K / V input of every decoder layer
                _need_fp32(orig)
                pl = make_planes(x3d.view(-1, x3d.shape[-1]), act_fmt(prec), pack=pack_of(orig))
                attach_planes(orig, pl)
            if pl.pack is not pack_of(orig):
                raise RuntimeError("MHAFn: the operand planes attached to an input are in another row layout than the input")
            return pl
        Qp = split(Q, Qc, prec_q)
        Kp = Qp if same_qk else split(K, Kc, prec_kv)
        Vp = Kp if same_kv else split(V, Vc, prec_kv)
        fuse = None
        q = k = v = None
        if same_qk and same_kv:
            r = project_group(Qp, (Wq, Wk, Wv), (bq, bk, bv), prec_q, qkv_fmt)
            if r is not None:
                (q, k, v), fuse = r, "qkv"
        elif same_kv:
            r = project_group(Kp, (Wk, Wv), (bk, bv), prec_kv, qkv_fmt)
            if r is not None:
                (k, v), fuse = r, "kv"
        if q is None:
            q = linear_fwd_planes(Qp, Wq, bq, precision=prec_q, out_fmt=qkv_fmt)
        if k is None:
            k = linear_fwd_planes(Kp, Wk, bk, precision=prec_kv, out_fmt=qkv_fmt)
            v = linear_fwd_planes(Vp, Wv, bv, precision=prec_kv, out_fmt=qkv_fmt)
        o, lse = attn_fwd_planes(q, k, v, B, Sq, Sk, D, mask, H, drop_p=p, site=site, precision=pol.attn, out_fmt=act_fmt(pol.gemm))
        epi = {}
        if res is not None:              # x_res + dropout(out-projection) in the projection's epilogue (ResidualConnection)
            r2 = _f32c(res).view(-1, Dq)
            epi = dict(residual=r2, ldr=r2.stride(0), drop_post=True, drop_p=res_p, site=res_site)
        opl = None
        if out_fmt is not None and Dq % 64 == 0:      # the reader of this result (the other modality's cross-attention) wants it as planes
            opl = _alloc_planes(B * Sq, Dq, out_fmt, Qc.device)
            epi["out_planes"] = opl
        out = linear_fwd(o, Wo, bo, precision=pol.gemm, **epi).view(B, Sq, Dq)
        carry_pack(qpack, out)
        if opl is not None:
            attach_planes(out, opl)
        if res is not None:
            request_grad_plane(out, res_p, res_site)
        ctx.H, ctx.p, ctx.site = H, p, site
        ctx.packs = (qpack, kpack)
        ctx.res = (res is not None, res_p, res_site)
        ctx.same_qk, ctx.same_kv = same_qk, same_kv
        ctx.fuse = fuse
        ctx.mask = mask
        ctx.dims = (B, Sq, Sk, D, Dq, Kc.shape[-1], Vc.shape[-1])
        ctx.params = (Wq, bq, Wk, bk, Wv, bv, Wo, bo)
        none = torch.empty(0, device=Qc.device)
        train = any(ctx.needs_input_grad)
        osec = o.lo if o.lo is not None else (o.fh if o.fh is not None else none)      # delta = rowsum(dO * O) reads hi + lo, or fp16
        ctx.o_f16 = o.fh is not None
        ctx.qkv_f16 = q.hi is None
        ctx.save_for_backward(Wq, Wk, Wv, Wo, q.any, k.any, v.any, o.hi, osec, lse,
                              Qp.hi if train else none, Kp.hi if train else none, Vp.hi if train else none)
        return out

    @staticmethod
    def backward(ctx, dout):
        Wq, Wk, Wv, Wo, qh, kh, vh, oh, osec, lse, QTh, KTh, VTh = ctx.saved_tensors
        B, Sq, Sk, D, Dq, Dk_in, Dv_in = ctx.dims
        Mq, Mk = B * Sq, B * Sk
        Wqp, bqp, Wkp, bkp, Wvp, bvp, Wop, bop = ctx.params
        qpack, kpack = ctx.packs
        if ctx.qkv_f16:
            q, k, v = Planes(None, None, Mq, D, fh=qh, pack=qpack), Planes(None, None, Mk, D, fh=kh, pack=kpack), Planes(None, None, Mk, D, fh=vh, pack=kpack)
        else:
            q, k, v = Planes(qh, None, Mq, D, pack=qpack), Planes(kh, None, Mk, D, pack=kpack), Planes(vh, None, Mk, D, pack=kpack)
        osec = osec if osec.numel() else None
        o = Planes(oh, None if ctx.o_f16 else osec, Mq, D, fh=osec if ctx.o_f16 else None, pack=qpack)
        # the inputs' own bf16 planes are the (k-major) dW operands
        QT, KT, VT = Planes(QTh, None, Mq, Dq, pack=qpack), Planes(KTh, None, Mk, Dk_in, pack=kpack), Planes(VTh, None, Mk, Dv_in, pack=kpack)
        dy2 = _f32c(dout).view(-1, Dq)
        has_res, res_p, res_site = ctx.res
        drop = None
        if has_res:                      # the residual branch takes dout as it is; this branch sees it through the dropout mask
            dy2, drop = drop_grad(dy2, bop, res_p, res_site)
        # out-projection: the dX epilogue re-applies the attention-output dropout mask -> gradient w.r.t. the pre-dropout output
        if D % 64 == 0:                  # dO is only ever an MFMA operand: bf16 plane, no fp32 copy
            P_, bias_done = grad_planes_from(dout, dy2, bop, drop, pack=qpack) if has_res else grad_planes(dy2, bop, drop=drop, pack=qpack)
            do = linear_dx(P_, Wop, out_planes=Planes(torch.empty(Mq, D, device=dy2.device, dtype=torch.bfloat16), None, Mq, D),
                           drop_post=True, drop_p=ctx.p, site=ctx.site)
            dWo, dbo = wgrad(Wop, None if bias_done else bop, P_, o, dy2_for_bias=dy2)
        else:
            do, dWo, dbo = lin_bwd(dy2, Wop, bop, o, drop=drop, pack=qpack, drop_post=True, drop_p=ctx.p, site=ctx.site)
        res = attn_bwd_planes(q, k, v, o, do, lse, B, Sq, Sk, D, ctx.mask, ctx.H, ctx.p, (bqp, bkp, bvp), fuse=ctx.fuse)
        (Pq, dbq), (Pk, dbk), (Pv, dbv) = res[:3]
        comb = res[3] if len(res) > 3 else None
        needQ, needK, needV = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        dQ = dK = dV = dWq = dWk = dWv = None

        def fused_bwd(Ws, xT, need_dx):
            """dX = [dq|dk|dv] . [Wq;Wk;Wv] as one GEMM; dW as one GEMM when the gradients are adjacent, else one per weight"""
            gst, _ = weight_group(Ws, tuple(None for _ in Ws), "bwd")
            dx = None
            if need_dx:          # [Wq;Wk;Wv] as stored ([3D][d_in]): its row is the reduction index
                dx = torch.empty(comb.rows, gst.cols, device=dy2.device, dtype=torch.float32)
                gT = weight_group_t(Ws) if comb.rows * gst.cols <= SMALL_DX_OUTPUTS else None
                if gT is not None and comb.hi.shape[1] == gT.hi.shape[1]:      # small: row-major on the 32 x 32 tile kernel
                    gemm_bf16(comb, gT, dx, ldc=dx.stride(0), precision=PREC_BF16)
                else:
                    gemm_bf16(comb, gst, dx, ldc=dx.stride(0), precision=PREC_BF16, b_km=True)
            gW = group_static_grad(Ws)
            if gW is not None:
                linear_dw(comb, xT, into=gW, params=Ws)
                for W in Ws:
                    grad_done(W)
                return dx, [None] * len(Ws)
            dWs, off = [], 0
            for W in Ws:
                N = W.shape[0]
                g1 = static_grad(W)
                dWs.append(linear_dw(Planes(comb.hi[:, off:off + N], None, comb.rows, N, pack=comb.pack), xT, into=g1, params=(W,)))
                if g1 is not None:
                    grad_done(W)
                off += N
            return dx, dWs

        if comb is not None and ctx.fuse == "qkv":
            dxq, (dWq, dWk, dWv) = fused_bwd((Wqp, Wkp, Wvp), QT, needQ)
            if needQ:
                dQ = dxq.view(B, Sq, Dq)
        elif comb is not None and ctx.fuse == "kv":
            dxq, dWq = lin_bwd_planes(Pq, Wqp, QT, need_dx=needQ)
            if needQ:
                dQ = dxq.view(B, Sq, Dq)
            need = needK or needV
            dxk, (dWk, dWv) = fused_bwd((Wkp, Wvp), KT, need)
            if need:
                dK = dxk.view(B, Sk, Dk_in)      # autograd adds dK and dV for the shared tensor; dV stays None
        else:
            dxq, dWq = lin_bwd_planes(Pq, Wqp, QT, need_dx=needQ)
            if ctx.same_qk and ctx.same_kv:     # one input, three contributions summed in the dX GEMM epilogue
                ldr = dxq.stride(0) if needQ else 0
                _, dWk = lin_bwd_planes(Pk, Wkp, KT, need_dx=needQ, out=dxq, residual=dxq, ldr=ldr)
                _, dWv = lin_bwd_planes(Pv, Wvp, VT, need_dx=needQ, out=dxq, residual=dxq, ldr=ldr)
                if needQ:
                    dQ = dxq.view(B, Sq, Dq)
            else:
                if needQ:
                    dQ = dxq.view(B, Sq, Dq)
                if ctx.same_kv:
                    need = needK or needV
                    dxk, dWk = lin_bwd_planes(Pk, Wkp, KT, need_dx=need)
                    _, dWv = lin_bwd_planes(Pv, Wvp, VT, need_dx=need, out=dxk, residual=dxk, ldr=dxk.stride(0) if need else 0)
                    if need:
                        dK = dxk.view(B, Sk, Dk_in)  # autograd adds dK and dV for the shared tensor; dV stays None
                else:
                    dxk, dWk = lin_bwd_planes(Pk, Wkp, KT, need_dx=needK)
                    dxv, dWv = lin_bwd_planes(Pv, Wvp, VT, need_dx=needV)
                    dK = dxk.view(B, Sk, Dk_in) if needK else None
                    dV = dxv.view(B, Sk, Dv_in) if needV else None
        return dQ, dK, dV, None, dWq, dbq, dWk, dbk, dWv, dbv, dWo, dbo, None, None, None, None, (dout if has_res else None), None, None, None


FUSE_GATE = True         # relu / dropout derivative inside the gradient's plane conversion (no fp32 dz tensor)
FUSE_GEN_LOSS = _os.environ.get("BMT_FUSE_GEN_LOSS", "1") != "0"      # switch: "0" = the generator and the loss as separate autograd nodes


class GenHandle:
    """what a Generator's output carries for a LabelSmoothing that is applied to it directly (K7 as one forward and one backward kernel):
    the generator's autograd-tracked input and parameters, the 2-D log-probabilities and their row sums"""
    __slots__ = ("x", "W", "b", "logp", "rowsum", "version", "ran", "xh")

    def __init__(self, x, W, b, logp, rowsum, xh=None):
        self.x, self.W, self.b, self.logp, self.rowsum = x, W, b, logp, rowsum
        self.version, self.ran = None, False
        self.xh = xh             # the bf16 plane of x the forward product read (k-major operand of dW: no second conversion in the backward)


def generator_handle(pred) -> Optional["GenHandle"]:
    """the GenHandle of ``pred`` if it is a Generator's untouched output (same values: tensor version unchanged)"""
    h = getattr(pred, "_bmt_gen", None)
    if h is None or h.version != pred._version or h.logp.numel() != pred.numel():
        return None
    return h


class GeneratorFn(torch.autograd.Function):
    """log_softmax(linear(x))   Generator.forward model/generators.py:18-19."""

    @staticmethod
    def forward(ctx, x, W, b):
        note_use(W, b)
        xc = _f32c(x)
        x2 = xc.view(-1, xc.shape[-1])
        V = W.shape[0]
        prec = policy_of(None).gemm
        xp = planes_of(x, act_fmt(prec))
        if xp is None or xp.rows != x2.shape[0] or xp.pack is not None:
            xp = make_planes(x2, act_fmt(prec))
        logp = linear_fwd(xp, W, b, precision=prec)
        # log-softmax with the row in registers (one read, one write) + the rows' sums, which are all LabelSmoothing needs of the tensor
        rowsum = torch.empty(logp.shape[0], device=logp.device, dtype=torch.float32)
        _lib.check(lib.bmt_log_softmax_fwd_stats(_p(logp), logp.stride(0), logp.shape[0], V, _p(rowsum), _st()), "bmt_log_softmax_fwd_stats")
        ctx.save_for_backward(x2, W, logp)
        ctx.params = (W, b)
        ctx.handle = GenHandle(x, W, b, logp, rowsum, xh=xp.hi)
        context().last_gen = ctx.handle
        return logp.view(*xc.shape[:-1], V)

    @staticmethod
    def backward(ctx, dlogp):
        ctx.handle.ran = True
        x2, W, logp = ctx.saved_tensors
        V = W.shape[0]
        d2 = _f32c(dlogp).view(-1, V)
        dlogits = torch.empty_like(d2)
        _lib.check(lib.bmt_log_softmax_bwd(_p(logp), logp.stride(0), _p(d2), d2.stride(0), _p(dlogits), dlogits.stride(0),
                                           d2.shape[0], V, _st()), "bmt_log_softmax_bwd")
        Wp, bp = ctx.params
        dx, dW, db = lin_bwd(dlogits, Wp, bp, x2, need_dx=ctx.needs_input_grad[0])
        if dx is not None:
            dx = dx.view(*dlogp.shape[:-1], W.shape[1])
        return dx, dW, db


def generator(x, W, b):
    """Generator.forward: the log-probabilities, carrying a GenHandle for a loss applied to them directly"""
    out = GeneratorFn.apply(x, W, b)
    c = context()
    h, c.last_gen = c.last_gen, None
    if h is not None and isinstance(out, torch.Tensor):
        h.version = out._version
        out._bmt_gen = h
    return out


def _ls_kl_forward(p2, t, smoothing, pad_idx, rowsum=None):
    """(loss 0-dim, row workspace) of LabelSmoothing.forward over 2-D log-probabilities; with the rows' sums in ONE launch"""
    rows, V = p2.shape
    loss = torch.empty((), device=p2.device, dtype=torch.float32)
    ws = torch.empty(rows + 1, device=p2.device, dtype=torch.float32)
    if rowsum is not None:
        _lib.check(lib.bmt_ls_kl_fwd_stats(_p(p2), p2.stride(0), _p(t), _p(rowsum), _p(loss), _p(ws), rows, V, smoothing, pad_idx, _st()),
                   "bmt_ls_kl_fwd_stats")
    else:
        _lib.check(lib.bmt_ls_kl_fwd(_p(p2), p2.stride(0), _p(t), _p(loss), _p(ws), rows, V, smoothing, pad_idx, _st()), "bmt_ls_kl_fwd")
    return loss, ws


class FusedGenLossFn(torch.autograd.Function):
    """LabelSmoothing(Generator(x)) as ONE autograd node over the generator's input and parameters (model/generators.py:18-19 +
    loss/label_smoothing.py:12-32): the forward reads the log-probabilities the generator already wrote (row sums + two gathers per row),
    the backward goes from them straight to the bf16 operand plane of d loss / d logits and its column sums (bmt_gen_lskl_bwd) and on to
    dX / dW -- the (B*Tc, V) tensor is read once and written once as 16 bits, instead of ls_kl_bwd -> log_softmax_bwd -> plane conversion.
    The gradient bypasses the log-probability tensor's own node (GeneratorFn), which still serves any OTHER consumer of the tensor."""

    @staticmethod
    def forward(ctx, x, W, b, target, smoothing, pad_idx, handle):
        note_use(W, b)
        logp = handle.logp
        t = target.contiguous().view(-1).long()
        loss, ws = _ls_kl_forward(logp, t, smoothing, pad_idx, handle.rowsum)
        xc = _f32c(x)
        ctx.save_for_backward(t, ws, logp, W, xc.view(-1, xc.shape[-1]))
        ctx.meta = (smoothing, pad_idx, x.shape)
        ctx.params = (W, b)
        ctx.handle = handle
        hs = context().gen_handles
        if len(hs) >= 8:          # (a caller that never settles them -- plain loss.backward() loops -- must not accumulate handles: ADVICE r4)
            del hs[:-7]
        hs.append(handle)
        return loss

    @staticmethod
    def backward(ctx, g):
        t, ws, logp, W, x2 = ctx.saved_tensors
        smoothing, pad_idx, xshape = ctx.meta
        Wp, bp = ctx.params
        rows, V = logp.shape
        gs = _f32c(g).reshape(1)
        P = Planes(torch.empty(rows, _pad64(V), device=logp.device, dtype=torch.bfloat16), None, rows, V)
        gb = static_grad(bp)
        cs = gb if gb is not None else (torch.zeros(V, device=logp.device, dtype=torch.float32) if bp is not None else None)
        _lib.check(lib.bmt_gen_lskl_bwd(_p(logp), logp.stride(0), _p(t), _p(ws), _p(gs), rows, V, smoothing, pad_idx, _p(P.hi), P.hi.stride(0),
                                        _p(cs), _st()), "bmt_gen_lskl_bwd")
        if gb is not None:
            grad_done(bp)
        dx = linear_dx(P, Wp).view(xshape) if ctx.needs_input_grad[0] else None
        h = ctx.handle
        xT = Planes(h.xh, None, x2.shape[0], x2.shape[1]) if (h.xh is not None and h.xh.shape[0] == x2.shape[0]) else x2
        dW, _ = wgrad(Wp, None, P, bwd_planes(xT))
        # this pass is done with the log-probabilities: the handle keeps the parameters only (38 MB at configs[1])
        h.x = h.logp = h.rowsum = h.xh = None
        return dx, dW, (None if gb is not None else cs), None, None, None, None


def label_smoothing(pred, target, smoothing: float, pad_idx: int):
    """LabelSmoothing.forward; on a Generator's untouched output (training) the fused node above"""
    h = generator_handle(pred) if FUSE_GEN_LOSS else None
    if h is not None and torch.is_grad_enabled() and pred.requires_grad and pred.is_cuda and h.logp.shape[1] > 2:
        return FusedGenLossFn.apply(h.x, h.W, h.b, target, smoothing, pad_idx, h)
    return LabelSmoothingFn.apply(pred, target, smoothing, pad_idx)


class LabelSmoothingFn(torch.autograd.Function):
    """LabelSmoothing.forward loss/label_smoothing.py:12-32 (sum-KL incl. the flat-index-0 pad quirk)."""

    @staticmethod
    def forward(ctx, pred, target, smoothing, pad_idx):
        V = pred.shape[-1]
        p2 = _f32c(pred).view(-1, V)
        t = target.contiguous().view(-1).long()
        h = generator_handle(pred)
        loss, ws = _ls_kl_forward(p2, t, smoothing, pad_idx, h.rowsum if h is not None else None)
        ctx.save_for_backward(t, ws)
        ctx.shape, ctx.smoothing, ctx.pad_idx = pred.shape, smoothing, pad_idx
        return loss

    @staticmethod
    def backward(ctx, g):
        t, ws = ctx.saved_tensors
        V = ctx.shape[-1]
        rows = t.numel()
        dpred = torch.empty(rows, V, device=t.device, dtype=torch.float32)
        gs = _f32c(g).reshape(1)
        _lib.check(lib.bmt_ls_kl_bwd(_p(t), _p(dpred), V, _p(gs), _p(ws), rows, V, ctx.smoothing, ctx.pad_idx, _st()),
                   "bmt_ls_kl_bwd")
        return dpred.view(ctx.shape), None, None, None


class PrepFeaturesFn(torch.autograd.Function):
    """dropout((a [+ b]) + PE)   model/captioning_module.py:165,174-175 + model/blocks.py:101-107."""

    @staticmethod
    def forward(ctx, a, b, pe, p, site, pack=None):
        ac = _f32c(a)
        bc = None if b is None else _f32c(b)
        B, S, D = ac.shape
        out = torch.empty_like(ac)
        ctx.packed = pack is not None
        if pack is not None:          # the valid rows only, compacted (RowPack); the features are data, nothing flows back
            _check_pack(pack, B * S)
            if any(ctx.needs_input_grad[:2]):
                raise RuntimeError("packed rows: feature stacks that require gradients (run with ops.PACK_ROWS = False)")
            _lib.check(lib.bmt_prep_features_packed(_p(ac), _p(bc), _p(pe), _p(out), B, S, D, p, _p(rng_tensor()) if p > 0 else None, site,
                                                    _p(pack.row_map), pack.rows_ptr, _st()), "bmt_prep_features_packed")
            carry_pack(pack, out)
        else:
            _lib.check(lib.bmt_prep_features(_p(ac), _p(bc), _p(pe), _p(out), B, S, D, p, _p(rng_tensor()) if p > 0 else None, site,
                                             _st()), "bmt_prep_features")
        ctx.p, ctx.site, ctx.has_b = p, site, b is not None
        return out

    @staticmethod
    def backward(ctx, dy):
        if ctx.packed:
            return None, None, None, None, None, None
        d = dropout_raw(_f32c(dy), ctx.p, ctx.site) if ctx.p > 0 else dy
        return d, (d if ctx.has_b else None), None, None, None, None


class EmbedFn(torch.autograd.Function):
    """dropout(W[ids] * sqrt(d) + PE)   model/blocks.py:42-46 + positional encoder."""

    @staticmethod
    def forward(ctx, ids, W, pe, scale, p, site):
        idc = ids.contiguous().long()
        B, S = idc.shape
        V, D = W.shape
        out = torch.empty(B, S, D, device=W.device, dtype=torch.float32)
        _lib.check(lib.bmt_prep_embed(_p(idc), _p(W), _p(pe), _p(out), B, S, D, V, scale, p, _p(rng_tensor()) if p > 0 else None,
                                      site, _st()), "bmt_prep_embed")
        ctx.save_for_backward(idc)
        ctx.meta = (V, D, scale, p, site)
        ctx.weight = W
        if W.requires_grad:
            note_use(W)
        return out

    @staticmethod
    def backward(ctx, dy):
        if not ctx.needs_input_grad[1]:
            return None, None, None, None, None, None
        (idc,) = ctx.saved_tensors
        V, D, scale, p, site = ctx.meta
        B, S = idc.shape
        gW = static_grad(ctx.weight)
        if gW is not None and (gW.shape != (V, D) or gW.dtype != torch.float32):
            gW = None
        # (the kernel scatters with atomics: a static gradient buffer takes them directly -- no zero-filled temporary, no accumulation pass)
        dW = gW if gW is not None else torch.zeros(V, D, device=dy.device, dtype=torch.float32)
        dyc = _f32c(dy)
        _lib.check(lib.bmt_prep_embed_bwd(_p(idc), _p(dyc), _p(dW), B, S, D, V, scale, p, _p(rng_tensor()) if p > 0 else None, site,
                                          _st()), "bmt_prep_embed_bwd")
        if gW is not None:
            grad_done(ctx.weight)
            return None, None, None, None, None, None
        return None, dW, None, None, None, None


# ----------------------------------------------------------------------------- fan-in / fan-out of the autograd graph as library launches
class Cat2Fn(torch.autograd.Function):
    """torch.cat([a, b], dim=-1) of two (..., D) fp32 tensors (the decoder layer's Ca | Cv in front of the bridge, model/decoders.py:83): one
    launch forward, one backward -- the two halves of the gradient leave as CONTIGUOUS tensors (autograd's narrow() views went through four
    ``.contiguous()`` copies per layer on their way into the LayerNorm / plane-conversion kernels)"""

    @staticmethod
    def forward(ctx, a, b):
        ac, bc = _f32c(a), _f32c(b)
        Da, Db = ac.shape[-1], bc.shape[-1]
        rows = ac.numel() // Da
        out = torch.empty(*ac.shape[:-1], Da + Db, device=ac.device, dtype=torch.float32)
        _lib.check(lib.bmt_cat2(_p(ac), Da, Da, _p(bc), Db, Db, _p(out), Da + Db, rows, _st()), "bmt_cat2")
        ctx.dims = (ac.shape, bc.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        sa, sb = ctx.dims
        gc = _f32c(g)
        Da, Db = sa[-1], sb[-1]
        ga = torch.empty(sa, device=g.device, dtype=torch.float32)
        gb = torch.empty(sb, device=g.device, dtype=torch.float32)
        _lib.check(lib.bmt_split2(_p(gc), Da + Db, _p(ga), Da, Da, _p(gb), Db, Db, ga.numel() // Da, _st()), "bmt_split2")
        return ga, gb


def cat2(a, b):
    if a.is_cuda and b.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32 and a.shape[:-1] == b.shape[:-1]:
        return Cat2Fn.apply(a, b)
    return torch.cat([a, b], dim=-1)


# ----------------------------------------------------------------------------- cross-attention against the RAW encoder memory
# model/multihead_attention.py:62-84 projects the encoder memory X (6 242 + 19 508 valid rows at configs[1]) to keys and values in every
# decoder layer -- the only large products of the decoder, their dX and a share of the weight-gradient launch behind them -- for 29 queries
# per sample.  With so few queries the products reassociate:
#     S_h = q_h K_h^T = (q_h W_k,h) X^T   (+ q_h . b_k: constant along the keys, the softmax does not see it)
#     O_h = P_h V_h   = (P_h X) W_v,h^T + b_v
# so the attention runs against X itself (one key / value plane for all heads, width d_memory) and K, V, dK, dV never exist.  Everything is
# a small product on the 32 x 32 tile kernel (bmt_gemm_small_batched, batch = (sample, head)), with the softmax as its own pass between
# them; the memory's gradient, sum over layers and heads of dS^T Q' + P^T dO', is ONE product per sample at the end of the decoder's
# backward (reduction over (layer, kind, head, query): the two operand stacks are laid out for it).  Operand formats (the study of
# tests/study_cross_attention_reassociation.py): the per-head block products q_h W_k,h and (P_h X) W_v,h^T split-bf16 (three passes), the two
# products against the memory one fp16 pass, backward one bf16 pass.  RAW_MEMORY switch: "0" = keys and values are projected, as before.
RAW_MEMORY = _os.environ.get("BMT_RAW_MEMORY", "1") != "0"
# RAW_FUSED: the two products against the memory and the row operation between them as ONE launch per attention (bmt_raw_attn_fwd / _bwd,
# ABI 12); False = the three launches of round 5 (the same arithmetic: tests/test_gpu_raw_memory.py compares the two).  RAW_FUSED_EDGES: ... and
# the block products either side of it (Q'_h = q_h W_k,h in front of the forward; dO'_h = do_h W_v,h in front of, dq_h = dQ'_h W_k,h^T behind the
# backward: bmt_raw_attn_fwd_edges / _bwd_edges).  Module attributes, not environment switches (tools/gpu_ab_attr.sh sets them for a same-box A/B:
# profiles/r06_z6_raw_fused_ab.txt).
RAW_FUSED = True
RAW_FUSED_EDGES = True
# ... and the stream-side projections in front of those: the forward's (q_h = y W_q,h^T + b_q,h inside the launch, bmt_raw_attn_fwd_proj: -0.02 ... -0.04
# ms/step) and the backward's (do_h = mask(dy W_o) inside the launch, bmt_raw_attn_bwd_proj: nothing measurable in the step, 4 launches and 4 MB fewer).  The
# forward's adds in the order of the launch it replaces -- per 16 reduction indices lo . hi, hi . lo, hi . hi into one accumulator -- and hands on the same
# bits (tests/test_gpu_raw_memory.py::test_query_projection_inside_the_launch_changes_no_bit).  Its first form (one staged chunk of W_q's high plane feeding
# both planes of y, the low plane in a second pass: fewer LDS bytes) was as close to the oracle as this one in every parity number and put the ten-Adam-step
# trajectory of tests/test_gpu_model.py -- chaotic in the gradients' low bits -- on another path, 2.6e-3 ... 3.1e-3 off the oracle's at step 5 against a
# 3e-3 bar (2e-4 ... 9e-4 before and now): profiles/r06_z6_raw_fused_ab.txt, tools/probes/adam_gap_arms.sh.
RAW_FUSED_PROJ = True
RAW_FUSED_PROJ_BWD = True


class RawMemoryState:
    """what the decoder layers share about ONE encoder memory during a step: its packed planes, the transposed per-sample copies, and the two
    operand stacks of the memory's gradient -- A [B][L][2][H][32][Skp] (kind 0: dS, kind 1: P), Bk [B][L][2][H][32][dm] (kind 0: Q' = q W_k,
    kind 1: dO'), bf16, rows t >= Tq zero."""
    __slots__ = ("pack", "x", "B", "S", "dm", "L", "H", "Tq", "Skp", "xt_f16", "xtc_bf", "astack", "bstack", "next_layer", "events", "used", "done")

    def a_block(self, l, kind):          # element offset of (b = 0, l, kind, h = 0) in the A stack; strides (sample, head)
        return ((l * 2 + kind) * self.H) * 32 * self.Skp, self.L * 2 * self.H * 32 * self.Skp, 32 * self.Skp

    def b_block(self, l, kind):
        return ((l * 2 + kind) * self.H) * 32 * self.dm, self.L * 2 * self.H * 32 * self.dm, 32 * self.dm

    def tensors(self):
        return [t for t in (self.xt_f16, self.xtc_bf, self.astack, self.bstack, self.x.hi, self.x.fh, self.pack.off, self.pack.row_map) if t is not None]


def _addr(t: torch.Tensor, elems: int = 0) -> int:
    return t.data_ptr() + elems * t.element_size()


def gemm_batched(prec, M, N, Kpad, nb_o, nb_i, ah, al, lda, bh, bl, ldb, *, a_off=(0, 0), b_off=(0, 0), b_rows=None, C_=None, ldc=0, c_off=(0, 0),
                 p1=None, p2=None, p2_f16=False, ldp=0, p_off=(0, 0), ldp2=0, p2_off=None, bias=None, bias_off_i=0, colsum=None, alpha=1.0,
                 drop_p=0.0, site=0, drop_off=(0, 0), a_div=(0, 0), c_div=(0, 0), p_div=(0, 0), p2_div=None, split=0):
    """nb_o x nb_i small products of one shape in one launch (bmt_gemm_small_batched).  ah / al / bh / bl / C_ / p1 / p2: ADDRESSES (ints;
    ``_addr``) of product (0, 0)'s operands and outputs; x_off = (per outer index, per inner index) element offsets; x_div = (rows per block, elements
    between blocks): row r of that operand / output lives at (r // rows) * elements + (r % rows) * ld (p2_div defaults to p_div)."""
    flags = (EPI_BIAS if bias is not None else 0) | (EPI_DROP_POST if drop_p > 0.0 else 0)
    a = GemmBf16Args(ah, al, lda, bh, bl, ldb, C_, ldc, p1, None if p2_f16 else p2, ldp, M, N, Kpad, alpha, flags,
                     _p(bias), None, 0, None, 0, 1.0, drop_p, _p(rng_tensor()) if drop_p > 0.0 else None, site, prec, split)
    if p2_f16:
        a.C_f16 = p2
    a.colsum = _p(colsum)
    p2_off = p_off if p2_off is None else p2_off
    p2_div = p_div if p2_div is None else p2_div
    bt = _lib.GemmBatch(nb_o, nb_i, a_off[0], a_off[1], b_off[0], b_off[1], b_rows, c_off[0], c_off[1], p_off[0], p_off[1], p2_off[0], p2_off[1], ldp2,
                        bias_off_i, drop_off[0], drop_off[1], a_div[0], c_div[0], p_div[0], p2_div[0], a_div[1], c_div[1], p_div[1], p2_div[1])
    _lib.check(lib.bmt_gemm_small_batched(C.byref(a), C.byref(bt), _st()), "bmt_gemm_small_batched")


def raw_attn_launch(bwd: bool, B: int, H: int, Tq: int, S: int, dm: int, fn, edges_dk: int = 0, proj_k: int = 0):
    """one fused launch of the reassociated cross-attention's middle (bmt_raw_attn_fwd / _bwd): two products of H Tq x S x dm per sample and the
    row operation between them (edges_dk: + the block products of H Tq x dm x d_k either side; proj_k: + the query projection H Tq x d_k x proj_k in front).  ``fn`` issues it --
    a seam of its own so that bench.py's kernel timer sees the launch as a class"""
    return fn()


def raw_form_ok(st: "RawMemoryState", Q, mha, pol) -> bool:
    """does this MultiheadedAttention call fit the reassociated form the state was prepared for?"""
    # (a state has one slot per decoder layer in its operand stacks: an attention call beyond them -- a layer re-executed by a checkpoint
    # recompute, a memory alias handed to an extra attention -- falls back to the projected form instead of writing past the stacks; and a
    # call that will need a backward at all -- the queries or ANY parameter of the module -- needs the state's gradient stacks)
    return (st is not None and st.next_layer < st.L and Q.dim() == 3 and Q.shape[0] == st.B and Q.shape[1] == st.Tq and mha.H == st.H and
            mha.d_model_K == st.dm and mha.d_model % mha.H == 0 and (mha.d_model // mha.H) % 64 == 0 and pol.gemm == PREC_BF16X3 and
            pol.attn == PREC_F16 and
            (not torch.is_grad_enabled() or st.astack is not None or not (Q.requires_grad or any(p_.requires_grad for p_ in mha.parameters()))))


def raw_memory(mem: torch.Tensor, n_layers: int, H: int, Tq: int, pol=None) -> torch.Tensor:
    """``mem`` (a packed encoder memory) as the decoder layers' key / value input for the reassociated cross-attention: an alias that carries the
    step's RawMemoryState, or ``mem`` itself where the form does not apply (not packed, more than 32 queries, inference, another operand
    policy than split-bf16 products around fp16 attention, switched off)"""
    pk = pack_of(mem)
    if pol is not None and not (pol.gemm == PREC_BF16X3 and pol.attn == PREC_F16):
        return mem
    if not torch.is_grad_enabled():      # inference projects keys and values: greedy decoding computes them once per caption (ops.mha_infer), and a
        return mem                       # full forward pass under no_grad runs the same kernels as that cached form (tests/test_gpu_model.py)
    if not (isinstance(mem, torch.Tensor) and mem.requires_grad):
        # a memory without a gradient (a frozen encoder: cfg.finetune_prop_encoder = False) has no operand stacks, and every training call
        # would fall back to the projected form (raw_form_ok): do not prepare transposed copies and a zeroed B stack nobody reads
        return mem
    if not (RAW_MEMORY and SMALL_DX_OUTPUTS > 0 and pk is not None and isinstance(mem, torch.Tensor) and mem.is_cuda and mem.dim() == 3 and
            mem.dtype == torch.float32 and mem.shape[-1] % 64 == 0 and 0 < Tq <= 32 and n_layers > 0 and mem.shape[1] <= 1024 and
            context().kv_cache is None):
        return mem
    B, S, dm = mem.shape
    x = planes_of(mem, "f16")
    if x is None:
        _need_fp32(mem)
        x = make_planes(_f32c(mem).view(-1, dm), "f16", pack=pk)
        attach_planes(mem, x)
    if x.pack is not pk:
        return mem
    st = RawMemoryState()
    st.pack, st.x, st.B, st.S, st.dm, st.L, st.H, st.Tq, st.Skp = pk, x, B, S, dm, n_layers, H, Tq, _pad64(S)
    st.next_layer, st.events, st.used, st.done = 0, [], False, set()
    train = mem.requires_grad and torch.is_grad_enabled()
    dev = mem.device
    st.xt_f16 = torch.empty(B, dm, st.Skp, device=dev, dtype=torch.float16)
    st.xtc_bf = torch.empty(B, dm, st.Skp, device=dev, dtype=torch.bfloat16) if train else None
    # one zeroed allocation: the B stack (rows t >= Tq are never written: finite zeros) and the workspace of the samples' key sums behind it
    nb_ = B * n_layers * 2 * H * 32 * dm
    raw = zero_(torch.empty(2 * nb_ + 4 * B * dm, device=dev, dtype=torch.uint8))
    st.bstack = raw[:2 * nb_].view(torch.bfloat16).view(B, n_layers, 2, H, 32, dm)
    ksum = raw[2 * nb_:].view(torch.float32)
    _lib.check(lib.bmt_memory_transposed(_p(x.fh), x.fh.stride(0), pk.off_ptr, B, dm, st.Skp, _p(st.xt_f16), _p(st.xtc_bf), _p(ksum), _st()),
               "bmt_memory_transposed")
    st.astack = torch.empty(B, n_layers, 2, H, 32, st.Skp, device=dev, dtype=torch.bfloat16) if train else None
    out = RawMemoryFn.apply(mem, st) if train else mem.view_as(mem)
    carry_pack(pk, out)
    attach_planes(out, x)
    out._bmt_rawmem = st
    return out


MEMGRAD_STORE = True      # the memory gradient's per-sample products stored instead of accumulated into a zeroed tensor (same-box A/B: tools/gpu_ab_attr.sh)


class RawMemoryFn(torch.autograd.Function):
    """memory -> its alias for the decoder layers; backward: the memory's gradient from the operand stacks the layers' backward passes filled,
    dX_b = sum over (layer, kind, head, query) of A_b[.]^T Bk_b[.] -- one product per sample, reduction 2 L H 32, written into the packed rows
    of sample b (bmt_gemm_bf16_grouped with the output's first row and row count in device memory)"""

    @staticmethod
    def forward(ctx, mem, st):
        ctx.st = st
        ctx.set_materialize_grads(False)
        return mem.view_as(mem)

    @staticmethod
    def backward(ctx, g):
        st = ctx.st
        cur = torch.cuda.current_stream()
        for ev in st.events:             # the layers' backward passes ran on their forward's streams and hand nothing over through autograd
            cur.wait_event(ev)
        st.events = []
        if not st.used:
            return g, None
        for t in st.tensors():
            t.record_stream(cur)
        B, S, dm, K = st.B, st.S, st.dm, st.L * 2 * st.H * 32
        for l in range(st.L):            # a layer that did not take this form (or whose backward did not run) left its rows of the A stack unwritten:
            if l not in st.done:         # zeros there (its rows of the B stack are zeros already; 0 x garbage must not be NaN)
                st.astack[:, l].zero_()
        # (a sample's product writes exactly its packed rows, each element once: stored, not accumulated -- no zero fill of the 33 / 10 MB in front
        # of it; rows past the packed count are never read: the consumers are bounded by the same device-side count)
        store = MEMGRAD_STORE and K <= 96 * 64
        dmem = torch.empty(B, S, dm, device=st.astack.device, dtype=torch.float32)
        if not store:
            zero_(dmem)
        A2, B2, out2 = st.astack.view(B, K, st.Skp), st.bstack.view(B, K, dm), dmem.view(B * S, dm)
        off = st.pack.off.data_ptr()
        items = [(Planes(A2[b], None, K, S), Planes(B2[b], None, K, dm), out2, (off + 4 * b, off + 4 * (B + 1 + b))) for b in range(B)]
        gemm_bf16_grouped(items, store=store)
        if g is not None:                # (a consumer outside the reassociated form read the alias too)
            a, b_ = _f32c(dmem), _f32c(g)
            _lib.check(lib.bmt_add(_p(a), _p(b_), _p(a), a.numel(), _st()), "bmt_add")
        return dmem, None


def _blockdiag_dw(dy: Planes, x: Planes, W: torch.Tensor, H: int):
    """dW[h] = dy[:, h-th column block]^T . x[:, h-th column block] for the H row blocks of W: into its static gradient buffer (returns None), or
    as a tensor"""
    gW = static_grad(W)
    tgt = gW if gW is not None else torch.zeros_like(W)
    nb, kb, M = W.shape[0] // H, x.cols // H, dy.rows
    for h in range(H):
        linear_dw(Planes(dy.hi[:, h * nb:(h + 1) * nb], None, M, nb, pack=dy.pack), Planes(x.hi[:, h * kb:(h + 1) * kb], None, M, kb, pack=x.pack),
                  into=tgt[h * nb:(h + 1) * nb], params=(W,) if gW is not None else ())
    if gW is not None:
        grad_done(W)
        return None
    return tgt


class RawCrossAttnFn(torch.autograd.Function):
    """MultiheadedAttention.forward (model/multihead_attention.py:55-86) for a decoder layer's attention over an encoder memory, in the
    reassociated form above.  Same arguments as MHAFn (K is V: the RawMemoryState-carrying alias of the memory)."""

    @staticmethod
    def forward(ctx, Q, K, V, mask, Wq, bq, Wk, bk, Wv, bv, Wo, bo, H, p, site, pol, res=None, res_p=0.0, res_site=0, out_fmt=None):
        st = K._bmt_rawmem
        note_use(Wq, bq, Wk, bk, Wv, bv, Wo, bo)
        Qc = _f32c(Q)
        B, Tq, Dq = Qc.shape
        D, dm, Skp, dk = Wq.shape[0], st.dm, st.Skp, Wq.shape[0] // H
        M = B * Tq
        dev = Qc.device
        l = st.next_layer
        st.next_layer += 1
        st.used = True
        cur = torch.cuda.current_stream()
        for t in st.tensors():
            t.record_stream(cur)
        X3 = PREC_BF16X3
        Qp = planes_of(Q, "x3")
        if Qp is None:
            _need_fp32(Q)
            Qp = make_planes(Qc.view(-1, Dq), "x3")
            attach_planes(Q, Qp)
        f_edges = RAW_FUSED and RAW_FUSED_EDGES and bool(lib.bmt_raw_attn_fwd_edges_ok(dm, Skp, dk))     # Q' inside the fused launch below (no fp16 copy in memory)
        Kq = Qp.hi.stride(0)
        # ... and the query projection in front of it: q_h = y W_q,h^T + b_q,h from the sample's rows of y (only its high plane reaches memory)
        f_proj = f_edges and RAW_FUSED_PROJ and Qp.lo is not None and Wq.dim() == 2 and bool(lib.bmt_raw_attn_fwd_proj_ok(dm, Skp, dk, Kq))
        if f_proj:
            wq = weight_planes(Wq, "x3")
            f_proj = wq.lo is not None and wq.hi.stride(0) >= Kq
        if f_proj:
            q = Planes(torch.empty(M, D, device=dev, dtype=torch.bfloat16), None, M, D)
        else:
            q = linear_fwd_planes(Qp, Wq, bq, precision=X3, out_fmt="x3")                     # [M][D] hi + lo
        # the memory's two projections as one weight group (what the projected form and greedy decoding register too): member planes
        grp, _ = weight_group((Wk, Wv), (bk, bv), "x3")
        gT = weight_group_t((Wk, Wv), lo=True, bs=(bk, bv))                                 # [dm][2 D] hi + lo: columns [0, D) = W_k^T
        train = any(ctx.needs_input_grad)      # (the queries, the memory or ANY of the module's parameters: a partly frozen module saves what its backward reads)
        # Q'[(b, t)][h dm + d] = q_h W_k,h: fp16 (the A operand of S) in the natural layout, bf16 into the B stack (b, l, 0, h)
        bo_, bsb, bsh = st.b_block(l, 0)
        if not f_edges:
            qf = torch.empty(M, H * dm, device=dev, dtype=torch.float16)
            # (one product per head over all the samples' rows: a weight tile is fetched once, not once per sample; row (b, t) -> block b of the stack)
            gemm_batched(X3, M, dm, D // H, 1, H, _addr(q.hi), _addr(q.lo), D, _addr(gT.hi), _addr(gT.lo), gT.hi.stride(0),
                         a_off=(0, dk), b_off=(0, dk), p1=_addr(st.bstack, bo_), ldp=dm, p_off=(0, bsh), p_div=(Tq, bsb), p2=_addr(qf), p2_f16=True,
                         ldp2=H * dm, p2_off=(0, dm), p2_div=(0, 0))
        Pf = torch.empty(B, H, 32, Skp, device=dev, dtype=torch.float16)
        ao, asb, ash = st.a_block(l, 1) if st.astack is not None else (0, 0, 0)
        p_bf = C.c_void_p(_addr(st.astack, ao)) if st.astack is not None else None
        # O' = P X (natural layout, split-bf16 planes: the A operand of the value block product)
        Op = _alloc_planes(M, H * dm, "x3", dev, ld=H * dm)
        if f_proj:
            # q_h = y W_q,h^T + b_q,h -> Q'_h = q_h W_k,h -> S = Q' X^T -> P = softmax -> O' = P X: one launch per attention, workgroup = (sample, head)
            raw_attn_launch(False, B, H, Tq, st.S, dm, lambda: _lib.check(lib.bmt_raw_attn_fwd_proj(
                _addr(Qp.hi), _addr(Qp.lo), Kq, Kq, _addr(wq.hi), _addr(wq.lo), wq.hi.stride(0), _p(bq), _addr(q.hi), D, _addr(gT.hi), _addr(gT.lo), gT.hi.stride(0),
                _addr(st.bstack, bo_), bsb, bsh, _addr(st.x.fh), st.x.fh.stride(0), st.pack.off_ptr, _addr(st.xt_f16), B, H, Tq, dm, Skp, dk, 1.0 / math.sqrt(dk),
                _p(Pf), p_bf, asb, ash, _addr(Op.hi), _addr(Op.lo), H * dm, _st()), "bmt_raw_attn_fwd_proj"), edges_dk=dk, proj_k=Dq)
        elif f_edges:
            # Q'_h = q_h W_k,h -> S = Q' X^T -> P = softmax -> O' = P X: one launch per attention, workgroup = (sample, head)
            raw_attn_launch(False, B, H, Tq, st.S, dm, lambda: _lib.check(lib.bmt_raw_attn_fwd_edges(
                _addr(q.hi), _addr(q.lo), D, _addr(gT.hi), _addr(gT.lo), gT.hi.stride(0), _addr(st.bstack, bo_), bsb, bsh, _addr(st.x.fh), st.x.fh.stride(0),
                st.pack.off_ptr, _addr(st.xt_f16), B, H, Tq, dm, Skp, dk, 1.0 / math.sqrt(dk), _p(Pf), p_bf, asb, ash, _addr(Op.hi), _addr(Op.lo), H * dm, _st()),
                "bmt_raw_attn_fwd_edges"), edges_dk=dk)
        elif RAW_FUSED and lib.bmt_raw_attn_ok(dm, Skp):
            # S = Q' X^T -> P = softmax -> O' = P X as one launch per attention, workgroup = (sample, head): the score tile stays in LDS
            raw_attn_launch(False, B, H, Tq, st.S, dm, lambda: _lib.check(lib.bmt_raw_attn_fwd(
                _addr(qf), Tq * H * dm, dm, H * dm, _addr(st.x.fh), st.x.fh.stride(0), st.pack.off_ptr, _addr(st.xt_f16), B, H, Tq, dm, Skp,
                1.0 / math.sqrt(dk), _p(Pf), p_bf, asb, ash, _addr(Op.hi), _addr(Op.lo), H * dm, _st()), "bmt_raw_attn_fwd"))
        else:
            # S = Q' X^T against the sample's packed rows
            S_ = torch.empty(B, H, 32, Skp, device=dev, dtype=torch.float32)
            # (one product per sample over the H Tq queries of all heads -- row (h, t): the memory's rows are fetched once for the four heads)
            gemm_batched(PREC_F16, H * Tq, st.S, dm, B, 1, _addr(qf), None, H * dm, _addr(st.x.fh), None, st.x.fh.stride(0),
                         a_off=(Tq * H * dm, 0), a_div=(Tq, dm), b_rows=st.pack.off_ptr, C_=_addr(S_), ldc=Skp, c_off=(H * 32 * Skp, 0), c_div=(Tq, 32 * Skp))
            _lib.check(lib.bmt_raw_softmax_fwd(_p(S_), st.pack.off_ptr, B, H, Tq, Skp, 1.0 / math.sqrt(dk), _p(Pf), p_bf, asb, ash, _st()), "bmt_raw_softmax_fwd")
            gemm_batched(PREC_F16, H * Tq, dm, Skp, B, 1, _addr(Pf), None, Skp, _addr(st.xt_f16), None, Skp,
                         a_off=(H * 32 * Skp, 0), a_div=(Tq, 32 * Skp), b_off=(dm * Skp, 0), p1=_addr(Op.hi), p2=_addr(Op.lo), ldp=H * dm,
                         p_off=(Tq * H * dm, 0), p_div=(Tq, dm))
        # concat_h(O'_h W_v,h^T + b_v), dropout on the attention output (model/multihead_attention.py:22-23), as split-bf16 planes
        o = _alloc_planes(M, D, "x3", dev, ld=D)
        gv_hi, gv_lo = grp.hi[D:], grp.lo[D:]                                               # W_v's rows of the group
        gemm_batched(X3, M, dk, dm, 1, H, _addr(Op.hi), _addr(Op.lo), H * dm, _addr(gv_hi), _addr(gv_lo), gv_hi.stride(0),
                     a_off=(0, dm), b_off=(0, dk * gv_hi.stride(0)), p1=_addr(o.hi), p2=_addr(o.lo), ldp=D, p_off=(0, dk), ldc=D,
                     bias=bv, bias_off_i=dk, drop_p=p, site=site, drop_off=(0, dk))
        epi = {}
        if res is not None:
            r2 = _f32c(res).view(-1, Dq)
            epi = dict(residual=r2, ldr=r2.stride(0), drop_post=True, drop_p=res_p, site=res_site)
        opl = None
        if out_fmt is not None and Dq % 64 == 0:
            opl = _alloc_planes(M, Dq, out_fmt, dev)
            epi["out_planes"] = opl
        out = linear_fwd(o, Wo, bo, precision=pol.gemm, **epi).view(B, Tq, Dq)
        if opl is not None:
            attach_planes(out, opl)
        if res is not None:
            request_grad_plane(out, res_p, res_site)
        ctx.st, ctx.l, ctx.H, ctx.p, ctx.site = st, l, H, p, site
        ctx.res = (res is not None, res_p, res_site)
        ctx.dims = (B, Tq, Dq, D)
        ctx.params = (Wq, bq, Wk, bk, Wv, bv, Wo, bo)
        none = torch.empty(0, device=dev)
        ctx.save_for_backward(Wq, Wk, Wv, Wo, q.hi if train else none, Pf if train else none, Op.hi if train else none, o.hi if train else none,
                              Qp.hi if train else none)
        return out

    @staticmethod
    def backward(ctx, dout):
        Wq_, Wk_, Wv_, Wo_, qh, Pf, Oph, oh, QTh = ctx.saved_tensors
        st, l, H, p = ctx.st, ctx.l, ctx.H, ctx.p
        B, Tq, Dq, D = ctx.dims
        Wq, bq, Wk, bk, Wv, bv, Wo, bo = ctx.params
        dm, Skp, dk, M = st.dm, st.Skp, D // H, B * Tq
        dev = dout.device
        cur = torch.cuda.current_stream()
        for t in st.tensors():
            t.record_stream(cur)
        dy2 = _f32c(dout).view(-1, Dq)
        has_res, res_p, res_site = ctx.res
        drop = None
        if has_res:
            dy2, drop = drop_grad(dy2, bo, res_p, res_site)
        # out-projection: dX with the attention-output dropout mask re-applied = gradient of concat_h(out_h); its column sums = db_v
        P_, bias_done = grad_planes_from(dout, dy2, bo, drop) if has_res else grad_planes(dy2, bo, drop=drop)
        gbv = static_grad(bv)
        dbv_t = gbv if gbv is not None else torch.zeros(D, device=dev, dtype=torch.float32)
        edges = RAW_FUSED and RAW_FUSED_EDGES and bool(lib.bmt_raw_attn_edges_ok(dm, Skp, dk))      # dO'_h and dq_h inside the fused launch below
        Kd = P_.hi.stride(0)
        # ... and this dX in front of them (do_h = mask(dy W_o[:, h-th block]) from the sample's rows of dy; the same transposed plane of W_o linear_dx reads)
        b_proj = (edges and RAW_FUSED_PROJ_BWD and M * D <= SMALL_DX_OUTPUTS and Wo.dim() == 2 and Wo.is_contiguous() and Wo.shape[0] <= 2048 and
                  bool(lib.bmt_raw_attn_bwd_proj_ok(dm, Skp, dk, Kd)))
        if b_proj:
            woT = weight_planes_t(Wo)
            b_proj = woT.hi.stride(0) >= Kd
        if b_proj:
            do = Planes(torch.empty(M, D, device=dev, dtype=torch.bfloat16), None, M, D)
        else:
            do = linear_dx(P_, Wo, out_planes=Planes(torch.empty(M, D, device=dev, dtype=torch.bfloat16), None, M, D), drop_post=True, drop_p=p, site=ctx.site,
                           colsum=dbv_t)
        if gbv is not None:
            grad_done(bv)
        dWo, dbo = wgrad(Wo, None if bias_done else bo, P_, Planes(oh, None, M, D), dy2_for_bias=dy2)
        grp, _ = weight_group((Wk, Wv), (bk, bv), "x3")
        gT = weight_group_t((Wk, Wv), lo=True, bs=(bk, bv))
        # dO'_h = do_h W_v,h  -> B stack (b, l, 1, h)
        bo_, bsb, bsh = st.b_block(l, 1)
        if not edges:
            gemm_batched(PREC_BF16, M, dm, dk, 1, H, _addr(do.hi), None, D, _addr(gT.hi, D), None, gT.hi.stride(0),
                         a_off=(0, dk), b_off=(0, dk), p1=_addr(st.bstack, bo_), ldp=dm, p_off=(0, bsh), p_div=(Tq, bsb))
        ao, asb, ash = st.a_block(l, 0)
        # dQ' = dS (X - mean key): natural layout, bf16
        dQp = Planes(torch.empty(M, H * dm, device=dev, dtype=torch.bfloat16), None, M, H * dm)
        gbq = static_grad(bq)
        dbq_t = gbq if gbq is not None else torch.zeros(D, device=dev, dtype=torch.float32)
        dq = Planes(torch.empty(M, D, device=dev, dtype=torch.bfloat16), None, M, D)
        if b_proj:
            # do_h = mask(dy W_o) (+ db_v) -> dO'_h = do_h W_v,h -> dP -> dS -> dQ' -> dq_h = dQ'_h W_k,h^T (+ db_q): one launch, workgroup = (sample, head)
            use_drop = p > 0.0
            raw_attn_launch(True, B, H, Tq, st.S, dm, lambda: _lib.check(lib.bmt_raw_attn_bwd_proj(
                _addr(P_.hi), Kd, Kd, _addr(woT.hi), woT.hi.stride(0), p if use_drop else 0.0, _p(rng_tensor()) if use_drop else None, ctx.site, _p(dbv_t),
                _addr(do.hi), D, _addr(gT.hi, D), gT.hi.stride(0), _addr(st.bstack, bo_), bsb, bsh, _addr(st.x.hi), st.x.hi.stride(0), st.pack.off_ptr,
                _addr(st.xtc_bf), _p(Pf), B, H, Tq, dm, Skp, dk, 1.0 / math.sqrt(dk), C.c_void_p(_addr(st.astack, ao)), asb, ash, _addr(dQp.hi), H * dm,
                _addr(grp.hi), grp.hi.stride(0), _addr(dq.hi), D, _p(dbq_t), _st()), "bmt_raw_attn_bwd_proj"), edges_dk=dk, proj_k=Dq)
        elif edges:
            # dO'_h = do_h W_v,h -> dP -> dS -> dQ' -> dq_h = dQ'_h W_k,h^T (+ column sums = db_q): one launch, workgroup = (sample, head)
            raw_attn_launch(True, B, H, Tq, st.S, dm, lambda: _lib.check(lib.bmt_raw_attn_bwd_edges(
                _addr(do.hi), D, _addr(gT.hi, D), gT.hi.stride(0), _addr(st.bstack, bo_), bsb, bsh, _addr(st.x.hi), st.x.hi.stride(0), st.pack.off_ptr,
                _addr(st.xtc_bf), _p(Pf), B, H, Tq, dm, Skp, dk, 1.0 / math.sqrt(dk), C.c_void_p(_addr(st.astack, ao)), asb, ash, _addr(dQp.hi), H * dm,
                _addr(grp.hi), grp.hi.stride(0), _addr(dq.hi), D, _p(dbq_t), _st()), "bmt_raw_attn_bwd_edges"), edges_dk=dk)
        elif RAW_FUSED and lib.bmt_raw_attn_ok(dm, Skp):
            # dP = dO' X^T -> dS = P o (dP - rowsum(P o dP)) scale -> dQ' = dS (X - mean key) as one launch (the forward's kernel, bf16 operands)
            raw_attn_launch(True, B, H, Tq, st.S, dm, lambda: _lib.check(lib.bmt_raw_attn_bwd(
                _addr(st.bstack, bo_), bsb, bsh, dm, _addr(st.x.hi), st.x.hi.stride(0), st.pack.off_ptr, _addr(st.xtc_bf), _p(Pf), B, H, Tq, dm, Skp,
                1.0 / math.sqrt(dk), C.c_void_p(_addr(st.astack, ao)), asb, ash, _addr(dQp.hi), H * dm, _st()), "bmt_raw_attn_bwd"))
        else:
            # dP = dO' X^T
            dP = torch.empty(B, H, 32, Skp, device=dev, dtype=torch.float32)
            gemm_batched(PREC_BF16, H * Tq, st.S, dm, B, 1, _addr(st.bstack, bo_), None, dm, _addr(st.x.hi), None, st.x.hi.stride(0),
                         a_off=(bsb, 0), a_div=(Tq, bsh), b_rows=st.pack.off_ptr, C_=_addr(dP), ldc=Skp, c_off=(H * 32 * Skp, 0), c_div=(Tq, 32 * Skp))
            _lib.check(lib.bmt_raw_softmax_bwd(_p(Pf), _p(dP), st.pack.off_ptr, B, H, Tq, Skp, 1.0 / math.sqrt(dk), C.c_void_p(_addr(st.astack, ao)), asb, ash, _st()),
                       "bmt_raw_softmax_bwd")
            gemm_batched(PREC_BF16, H * Tq, dm, Skp, B, 1, _addr(st.astack, ao), None, Skp, _addr(st.xtc_bf), None, Skp,
                         a_off=(asb, 0), a_div=(Tq, ash), b_off=(dm * Skp, 0), p1=_addr(dQp.hi), ldp=H * dm, p_off=(Tq * H * dm, 0), p_div=(Tq, dm))
        dWv = _blockdiag_dw(do, Planes(Oph, None, M, H * dm), Wv, H)      # (behind the launch that may have produced do)
        # dq_h = dQ'_h W_k,h^T (+ its column sums = db_q)
        if not edges:
            gemm_batched(PREC_BF16, M, dk, dm, 1, H, _addr(dQp.hi), None, H * dm, _addr(grp.hi), None, grp.hi.stride(0),
                         a_off=(0, dm), b_off=(0, dk * grp.hi.stride(0)), p1=_addr(dq.hi), ldp=D, p_off=(0, dk), colsum=dbq_t, bias_off_i=dk)
        if gbq is not None:
            grad_done(bq)
        dWk = _blockdiag_dw(Planes(qh, None, M, D), dQp, Wk, H)
        dbk = None
        if bk is not None:               # the key bias does not reach the output: its gradient is exactly zero
            if static_grad(bk) is not None:
                grad_done(bk)
            else:
                dbk = torch.zeros_like(bk)
        needQ = ctx.needs_input_grad[0]
        dxq, dWq = lin_bwd_planes(dq, Wq, Planes(QTh, None, M, Dq), need_dx=needQ)
        dQ = dxq.view(B, Tq, Dq) if needQ else None
        st.done.add(l)
        ev = torch.cuda.Event()
        ev.record(cur)
        st.events.append(ev)
        return (dQ, None, None, None, dWq, None if gbq is not None else dbq_t, dWk, dbk, dWv, None if gbv is not None else dbv_t, dWo, dbo,
                None, None, None, None, (dout if has_res else None), None, None, None)


# ---- the encoder's self-attention over an input NARROWER than a head (round 6) ---------------------------------------------------------
# model/multihead_attention.py:62-84 projects the audio stream (d_model_audio = 128) to d_model = 1024 for H = 4 heads of d_k = 256: q_h, k_h,
# v_h are rank-128 images of the same 128-wide LayerNorm output x.  The products reassociate exactly as the decoder's cross-attentions did
# in round 5 (RawCrossAttnFn) -- here with the input as its own memory:
#     S_h = q_h k_h^T = (x W_q,h^T + b_q,h)(x W_k,h^T + b_k,h)^T = (x A_h + b_q,h W_k,h) x^T + [constant along the keys: the softmax does not see it]
#                                                         A_h = W_q,h^T W_k,h  [d_in x d_in]
#     O_h = P_h v_h   = (P_h x) W_v,h^T + b_v,h
# so the attention runs with queries q' = x W'^T + c of H x d_in columns (W'_h = A_h^T = W_k,h^T W_q,h, c_h = b_q,h W_k,h) against ONE key /
# value plane of width d_in shared by the heads (kv_shared): both attention products at d_in instead of d_k, the 3 D-wide fused q / k / v
# projection (output-bound: 19 508 x 3072 planes at configs[1]) replaced by one H d_in-wide product, its dX by one [M][3 H d_in] x [3 H d_in][d_in]
# product (dq' through W', the per-head dK' / dV' summed by rows of identity blocks), and the value projection applied head by head to the
# H d_in-wide attention output (a block product on the reduction-of-128 kernel: bmt_gemm_bf16_args.a_blk_n; the attention-output dropout of
# :22-23 in its epilogue).  W' and c are functions of the weights alone: one fp32 kernel per optimizer step (bmt_rank_prep) writes them as
# operand planes; their gradients
#     dW_q,h = W_k,h dW'_h          dW_k,h = W_q,h dW'_h^T + b_q,h^T dc_h          db_q,h = W_k,h dc_h          (dW' = dq'^T x, dc = colsum dq')
# come from ONE item of the step's grouped weight-gradient launch (dW') and one fp32 kernel behind it (bmt_rank_chain).  The key bias drops
# out (its gradient is exactly zero, as the reference's is up to rounding).
RANK_ATTN = True
RANK_QUEUE_DC = True      # dc's partial sums in the pass's one deferred column-sum launch instead of a launch behind every rank-form attention backward
RANK_CROSS = True       # ... and a cross-attention over such an input (the video stream's attention over the audio stream): RankCrossAttnFn
_rank_states = {}


class _RankState:
    __slots__ = ("refs", "epoch", "c", "WpP", "Wcomb", "Istack", "dWp", "dc", "dc_used", "dirty", "H", "d_a", "d_b", "dk")


def _rank_prep(st: "_RankState") -> bool:
    """W' (operand planes) and c of one module from its weights as they are now, on the current stream (bmt_rank_prep)"""
    Wq, Wk, bq = (r() if r is not None else None for r in st.refs)
    if Wq is None or Wk is None:
        return False
    Wqd, Wkd = Wq.detach(), Wk.detach()
    _lib.check(lib.bmt_rank_prep(_p(Wqd), Wqd.stride(0), st.d_b, _p(Wkd), Wkd.stride(0), _p(bq.detach()) if bq is not None else None, st.H, st.dk, st.d_a,
                                 _p(st.WpP.hi), _p(st.WpP.fh), _p(st.WpP.fl), st.d_b, None, _p(st.c), _p(st.dWp), _st()), "bmt_rank_prep")
    st.epoch = WEIGHT_EPOCH[0]
    st.dirty = False                 # (the same launch zeroed dW', the accumulator of the pass to come)
    return True


def _rank_dc(st: "_RankState"):
    """the module's dc accumulator (column sums of dq') while it is clean: zeroed at creation and again by whoever consumed it -- the chain-rule
    launch's issuer, behind that launch (_rank_weight_grads): no fill in front of the attention backward, none on the refresh stream in front
    of the forward pass.  In use (a second backward before the first one's chain rule was issued): None = a fresh tensor."""
    if st.dc_used:
        return None
    st.dc_used = True
    return st.dc


def _rank_refresh_all():
    """called by the weight-plane registry's once-per-optimizer-step refresh (on its stream: beside the step's prologue, ops.EARLY_REFRESH)"""
    for k_, st in list(_rank_states.items()):
        if st.epoch != WEIGHT_EPOCH[0] and not _rank_prep(st):
            del _rank_states[k_]


def rank_form_ok(Q, K, V, mha, pol) -> bool:
    """does this MultiheadedAttention call take the rank form?  Keys = values = a CUDA input of 128 columns, at most half a head -- the queries
    the same tensor (self-attention) or another stream of a multiple of 128 columns --, under the encoder's operand policy"""
    if not (RANK_ATTN and K is V and isinstance(K, torch.Tensor) and K.is_cuda and K.dim() == 3 and isinstance(Q, torch.Tensor) and Q.dim() == 3):
        return False
    d_a, D, H = mha.d_model_K, mha.d_model, mha.H
    if Q is not K and not (RANK_CROSS and mha.d_model_Q % 128 == 0 and getattr(K, "_bmt_rawmem", None) is None):
        return False
    return (d_a == 128 and D % H == 0 and 2 * d_a <= D // H and (D // H) % 128 == 0 and mha.d_model_V == d_a and (Q is not K or mha.d_model_Q == d_a) and
            pol.gemm == PREC_F16W2 and pol.kv_gemm == PREC_F16W2 and pol.attn == PREC_F16 and QKV_F16_ONLY and context().kv_cache is None)


def _rank_state(Wq, bq, Wk, H) -> "_RankState":
    import weakref
    key = (id(Wq), id(Wk))
    st = _rank_states.get(key)
    D, d_b = Wq.shape
    d_a = Wk.shape[1]
    dk, Dr = D // H, H * d_a
    dev = Wq.device
    if st is None or any(r() is not w for r, w in zip(st.refs, (Wq, Wk))):
        st = _RankState()
        st.refs = [weakref.ref(Wq), weakref.ref(Wk), weakref.ref(bq) if bq is not None else None]
        st.H, st.d_a, st.d_b, st.dk, st.epoch = H, d_a, d_b, dk, -1
        st.c = torch.zeros(Dr, device=dev, dtype=torch.float32)
        eye = torch.eye(d_a, device=dev, dtype=torch.bfloat16).repeat(2 * H, 1)
        if d_b == d_a:
            # self-attention: W' (bf16) is the first H d_a rows of the combined dX weight [W' ; I-stack ; I-stack]
            st.Wcomb = torch.zeros(3 * Dr, d_a, device=dev, dtype=torch.bfloat16)
            st.Wcomb[Dr:] = eye
            hi, st.Istack = st.Wcomb[:Dr], None
        else:
            st.Wcomb, st.Istack = None, eye      # dx of the key / value input = sum_h (dK'_h + dV'_h): a product against the identity stack
            hi = torch.empty(Dr, d_b, device=dev, dtype=torch.bfloat16)
        # W' as the forward's two-plane operand (fh + fl) and, bf16, as the k-major operand of dy = dq' W'
        st.WpP = Planes(hi, None, Dr, d_b, fh=torch.empty(Dr, d_b, device=dev, dtype=torch.float16), fl=torch.empty(Dr, d_b, device=dev, dtype=torch.float16))
        # dW' of a pass that queues its weight-gradient products accumulates here: zeroed by bmt_rank_prep, i.e. once per optimizer step (dirty =
        # a pass has used it since: another one before the next refresh takes a buffer of its own)
        st.dWp = torch.zeros(Dr, d_b, device=dev, dtype=torch.float32)
        st.dc = torch.zeros(Dr, device=dev, dtype=torch.float32)
        st.dirty = st.dc_used = False
        _rank_states[key] = st
        if len(_rank_states) > 256:
            for k_ in [k_ for k_, v_ in _rank_states.items() if any(r is not None and r() is None for r in v_.refs)]:
                del _rank_states[k_]
    # once per optimizer step, with the refresh of the weights' operand planes (inside a captured step both are part of the graph); a module
    # the refresh did not know yet computes its own here
    with _weights.lock:
        _await_planes()
        _weights.ensure_fresh()
    if st.epoch != WEIGHT_EPOCH[0]:
        _rank_prep(st)
    return st


def _rank_deferred(st, Wq, bq, Wk) -> bool:
    """will this backward's dW' product and chain rule be queued for the pass's flush (static gradient buffers, first pass of the optimizer step)?"""
    n = sum(1 for p_ in (Wq, Wk, bq) if static_grad(p_) is not None)
    return context().defer_dw and n == (3 if bq is not None else 2) and not st.dirty


def _rank_weight_grads(st, Pq: Planes, yT: Planes, dc, Wq, bq, Wk):
    """dW' = dq'^T y into the pass's grouped weight-gradient launch, and behind it the chain rule through W' and c (bmt_rank_chain: fp32, from
    the parameters) into the gradients of W_q, b_q, W_k -- their static buffers, or fresh tensors (returned: (dW_q, db_q, dW_k), None = accumulated)"""
    gWq, gWk, gbq = static_grad(Wq), static_grad(Wk), static_grad(bq)
    tq = gWq if gWq is not None else torch.zeros_like(Wq)
    tk = gWk if gWk is not None else torch.zeros_like(Wk)
    tb = (gbq if gbq is not None else torch.zeros_like(bq)) if bq is not None else None
    static = [p_ for p_, g_ in ((Wq, gWq), (Wk, gWk), (bq, gbq)) if g_ is not None]
    sctx = context()
    # (gradients handed back to autograd as tensors must be complete now; so must a second pass over the module within one optimizer step)
    deferred = _rank_deferred(st, Wq, bq, Wk)
    acc = st.dWp if deferred else torch.zeros_like(st.dWp)
    Wqd, Wkd = Wq.detach(), Wk.detach()

    def chain():
        _lib.check(lib.bmt_rank_chain(_p(Wqd), Wqd.stride(0), st.d_b, _p(Wkd), Wkd.stride(0), _p(bq.detach()) if bq is not None else None, st.H, st.dk, st.d_a,
                                      _p(acc), _p(dc) if bq is not None else None, _p(tq), tq.stride(0), _p(tk), tk.stride(0), _p(tb), _st()), "bmt_rank_chain")
        if dc is st.dc:              # consumed: clean again for the next pass (same stream, behind the launch that read it)
            zero_(st.dc)
            st.dc_used = False
    if deferred:                     # dW' and the chain rule behind it run beside the pass's grouped weight-gradient launch (flush_dw)
        sctx.pending_side.append(((Pq, yT, acc), chain))
        sctx.pending_ids.update(id(p_) for p_ in static)
        st.dirty = True
    else:
        was, sctx.defer_dw = sctx.defer_dw, False
        try:
            linear_dw(Pq, yT, into=acc)
        finally:
            sctx.defer_dw = was
        chain()
    for p_ in static:
        grad_done(p_)
    return None if gWq is not None else tq, None if (gbq is not None or bq is None) else tb, None if gWk is not None else tk


def _value_planes(Wq, bq, Wk, bk, Wv, bv, fmt: str) -> Planes:
    """W_v's operand planes [D][d_a]: its rows of the module's projection group -- q | k | v of a self-attention, k | v of a cross-attention (Wq None):
    what the projected form registers, either form may meet the weights first --, or its own planes where projections are not grouped"""
    Ws, bs = ((Wq, Wk, Wv), (bq, bk, bv)) if Wq is not None else ((Wk, Wv), (bk, bv))
    grp = weight_group(Ws, bs, fmt)
    if grp is None:
        return weight_planes(Wv, fmt)
    g, D, r0 = grp[0], Wv.shape[0], (len(Ws) - 1) * Wv.shape[0]
    sl = lambda t: None if t is None else t[r0:r0 + D]
    return Planes(sl(g.hi), sl(g.lo), D, Wv.shape[1], fh=sl(g.fh), fl=sl(g.fl))


class RankSelfAttnFn(torch.autograd.Function):
    """MultiheadedAttention.forward (model/multihead_attention.py:55-86) for a self-attention over an input narrower than a head, in the
    reassociated form above.  Same arguments as MHAFn (Q is K is V)."""

    @staticmethod
    def forward(ctx, Q, K, V, mask, Wq, bq, Wk, bk, Wv, bv, Wo, bo, H, p, site, pol, res=None, res_p=0.0, res_site=0, out_fmt=None):
        note_use(Wq, bq, Wk, bk, Wv, bv, Wo, bo)
        Qc = _f32c(Q)
        B, S, d_in = Qc.shape
        D = Wq.shape[0]
        dk, M, Dr = D // H, B * S, H * d_in
        dev = Qc.device
        pack = _check_pack(pack_of(Q), M)
        xP = planes_of(Q, "f16")
        if xP is None:
            _need_fp32(Q)
            xP = make_planes(Qc.view(-1, d_in), "f16", pack=pack_of(Q))
            attach_planes(Q, xP)
        if xP.pack is not pack_of(Q):
            raise RuntimeError("RankSelfAttnFn: the operand planes attached to the input are in another row layout than the input")
        st = _rank_state(Wq, bq, Wk, H)
        # q' = x W'^T + c as the fp16 plane the attention kernels read
        qp = _alloc_planes(M, Dr, "f16only", dev, ld=Dr)
        qp.pack = pack
        gemm_bf16(xP, st.WpP, None, bias=st.c, out_planes=qp, precision=PREC_F16W2)
        kP = Planes(None, None, M, d_in, fh=xP.fh, pack=pack)
        scale = 1.0 / math.sqrt(dk)
        op, lse = attn_fwd_planes(qp, kP, kP, B, S, S, Dr, mask, H, precision=pol.attn, out_fmt="f16", kv_shared=True, scale=scale)
        # concat_h(O'_h W_v,h^T + b_v,h), dropout on the attention output (model/multihead_attention.py:22-23), as the out-projection's operand planes
        o = _alloc_planes(M, D, act_fmt(pol.gemm), dev, ld=D)
        o.pack = pack
        gemm_bf16(op, _value_planes(Wq, bq, Wk, bk, Wv, bv, weight_fmt(PREC_F16W2)), None, bias=bv, out_planes=o, precision=PREC_F16W2, drop_post=True,
                  drop_p=p, site=site, a_blk=(dk, d_in))
        epi = {}
        if res is not None:
            r2 = _f32c(res).view(-1, d_in)
            epi = dict(residual=r2, ldr=r2.stride(0), drop_post=True, drop_p=res_p, site=res_site)
        opl = None
        if out_fmt is not None and d_in % 64 == 0:
            opl = _alloc_planes(M, d_in, out_fmt, dev)
            epi["out_planes"] = opl
        out = linear_fwd(o, Wo, bo, precision=pol.gemm, **epi).view(B, S, d_in)
        carry_pack(pack, out)
        if opl is not None:
            attach_planes(out, opl)
        if res is not None:
            request_grad_plane(out, res_p, res_site)
        ctx.st, ctx.H, ctx.p, ctx.site, ctx.mask, ctx.pack = st, H, p, site, mask, pack
        ctx.res = (res is not None, res_p, res_site)
        ctx.dims = (B, S, d_in, D)
        ctx.params = (Wq, bq, Wk, bk, Wv, bv, Wo, bo)
        none = torch.empty(0, device=dev)
        train = any(ctx.needs_input_grad)
        ctx.save_for_backward(Wq, Wk, Wv, Wo, qp.fh, xP.fh, op.hi if train else none, op.fh if train else none, lse, o.hi if train else none,
                              xP.hi if train else none)
        return out

    @staticmethod
    def backward(ctx, dout):
        Wq_, Wk_, Wv_, Wo_, qf, xf, oph, opf, lse, oh, xh = ctx.saved_tensors
        st, H, p, pack = ctx.st, ctx.H, ctx.p, ctx.pack
        B, S, d_in, D = ctx.dims
        Wq, bq, Wk, bk, Wv, bv, Wo, bo = ctx.params
        dk, M, Dr = D // H, B * S, H * d_in
        dev = dout.device
        dy2 = _f32c(dout).view(-1, d_in)
        has_res, res_p, res_site = ctx.res
        drop = None
        if has_res:
            dy2, drop = drop_grad(dy2, bo, res_p, res_site)
        # out-projection: dX with the attention-output dropout mask re-applied = gradient of concat_h(O_h); its column sums = db_v
        P_, bias_done = grad_planes_from(dout, dy2, bo, drop, pack=pack) if has_res else grad_planes(dy2, bo, drop=drop, pack=pack)
        gbv = static_grad(bv)
        dbv_t = gbv if gbv is not None else (torch.zeros(D, device=dev, dtype=torch.float32) if bv is not None else None)
        do = linear_dx(P_, Wo, out_planes=Planes(torch.empty(M, D, device=dev, dtype=torch.bfloat16), None, M, D, pack=pack), drop_post=True, drop_p=p,
                       site=ctx.site, colsum=dbv_t)
        if gbv is not None:
            grad_done(bv)
        dWo, dbo = wgrad(Wo, None if bias_done else bo, P_, Planes(oh, None, M, D, pack=pack), dy2_for_bias=dy2)
        # dO'_h = dO_h W_v,h (a block product over W_v's bf16 plane, k-major: every head its own reduction rows); dW_v,h = dO_h^T O'_h
        dop = Planes(torch.empty(M, Dr, device=dev, dtype=torch.bfloat16), None, M, Dr, pack=pack)
        WvB = _value_planes(Wq, bq, Wk, bk, Wv, bv, "bwd")
        gemm_bf16(do, Planes(WvB.hi, None, D, d_in), None, out_planes=dop, precision=PREC_BF16, b_km=True, a_blk=(d_in, dk), splitk=1)
        oP = Planes(oph, None, M, Dr, fh=opf, pack=pack)
        dWv = _blockdiag_dw(do, oP, Wv, H)
        # attention backward against the shared plane: dq' | per-head dK' | per-head dV' in one [M][3 H d_in] plane, dc = column sums of dq'
        qp = Planes(None, None, M, Dr, fh=qf, pack=pack)
        kP = Planes(None, None, M, d_in, fh=xf, pack=pack)
        res = attn_bwd_planes(qp, kP, kP, oP, dop, lse, B, S, S, Dr, ctx.mask, H, 0.0, (st.c if bq is not None else None, None, None), fuse="qkv",
                              kv_shared=True, scale=1.0 / math.sqrt(dk), bias_into=(_rank_dc(st), None, None), queue_bias=RANK_QUEUE_DC and _rank_deferred(st, Wq, bq, Wk))
        (Pq, dc), comb = res[0], res[3]
        xT = Planes(xh, None, M, d_in, pack=pack)
        dQ = None
        if ctx.needs_input_grad[0]:      # dx = dq' W' + sum_h (dK'_h + dV'_h): one product against [W' ; I-stack ; I-stack]
            dx = torch.empty(M, d_in, device=dev, dtype=torch.float32)
            gemm_bf16(comb, Planes(st.Wcomb, None, 3 * Dr, d_in), dx, ldc=d_in, precision=PREC_BF16, b_km=True)
            dQ = dx.view(B, S, d_in)
        # dW' = dq'^T x joins the pass's grouped weight-gradient launch; the chain rule through W' and c runs behind it (fp32, from the parameters)
        dWq, dbq, dWk = _rank_weight_grads(st, Pq, xT, dc, Wq, bq, Wk)
        dbk = None
        if bk is not None:               # the key bias does not reach the output
            if static_grad(bk) is not None:
                grad_done(bk)
            else:
                dbk = torch.zeros_like(bk)
        return (dQ, None, None, None, dWq, dbq, dWk, dbk, dWv, None if gbv is not None else dbv_t, dWo, dbo, None, None, None, None,
                (dout if has_res else None), None, None, None)


class RankCrossAttnFn(torch.autograd.Function):
    """MultiheadedAttention.forward (model/multihead_attention.py:55-86) for an attention over keys = values narrower than a head from queries of
    another stream (the video stream's attention over the 128-wide audio stream, model/encoders.py:69-79), in the reassociated form above:
    q' = y W'^T + c with W'_h = W_k,h^T W_q,h [d_a][d_q] -- HALF the query projection's columns --, the attention at width d_a against the audio
    stream's own plane, no key / value projections.  Same arguments as MHAFn (K is V)."""

    @staticmethod
    def forward(ctx, Q, K, V, mask, Wq, bq, Wk, bk, Wv, bv, Wo, bo, H, p, site, pol, res=None, res_p=0.0, res_site=0, out_fmt=None):
        note_use(Wq, bq, Wk, bk, Wv, bv, Wo, bo)
        Qc, Kc = _f32c(Q), _f32c(K)
        B, Sq, Dq = Qc.shape
        Sk, d_a = Kc.shape[1], Kc.shape[2]
        D = Wq.shape[0]
        dk, Mq, Mk, Dr = D // H, B * Sq, B * Sk, H * d_a
        dev = Qc.device
        qpack, kpack = _check_pack(pack_of(Q), Mq), _check_pack(pack_of(K), Mk)

        def planes(orig, x3d, fmt, pk):
            pl = planes_of(orig, fmt)
            if pl is None:
                _need_fp32(orig)
                pl = make_planes(x3d.view(-1, x3d.shape[-1]), fmt, pack=pack_of(orig))
                attach_planes(orig, pl)
            if pl.pack is not pack_of(orig):
                raise RuntimeError("RankCrossAttnFn: the operand planes attached to an input are in another row layout than the input")
            return pl
        yP = planes(Q, Qc, act_fmt(pol.gemm), qpack)
        xP = planes(K, Kc, "f16", kpack)
        st = _rank_state(Wq, bq, Wk, H)
        qp = _alloc_planes(Mq, Dr, "f16only", dev, ld=Dr)
        qp.pack = qpack
        gemm_bf16(yP, st.WpP, None, bias=st.c, out_planes=qp, precision=PREC_F16W2)
        kP = Planes(None, None, Mk, d_a, fh=xP.fh, pack=kpack)
        scale = 1.0 / math.sqrt(dk)
        op, lse = attn_fwd_planes(qp, kP, kP, B, Sq, Sk, Dr, mask, H, precision=pol.attn, out_fmt="f16", kv_shared=True, scale=scale)
        o = _alloc_planes(Mq, D, act_fmt(pol.gemm), dev, ld=D)
        o.pack = qpack
        gemm_bf16(op, _value_planes(None, None, Wk, bk, Wv, bv, weight_fmt(PREC_F16W2)), None, bias=bv, out_planes=o, precision=PREC_F16W2, drop_post=True,
                  drop_p=p, site=site, a_blk=(dk, d_a))
        epi = {}
        if res is not None:
            r2 = _f32c(res).view(-1, Dq)
            epi = dict(residual=r2, ldr=r2.stride(0), drop_post=True, drop_p=res_p, site=res_site)
        opl = None
        if out_fmt is not None and Dq % 64 == 0:
            opl = _alloc_planes(Mq, Dq, out_fmt, dev)
            epi["out_planes"] = opl
        out = linear_fwd(o, Wo, bo, precision=pol.gemm, **epi).view(B, Sq, Dq)
        carry_pack(qpack, out)
        if opl is not None:
            attach_planes(out, opl)
        if res is not None:
            request_grad_plane(out, res_p, res_site)
        ctx.st, ctx.H, ctx.p, ctx.site, ctx.mask, ctx.packs = st, H, p, site, mask, (qpack, kpack)
        ctx.res = (res is not None, res_p, res_site)
        ctx.dims = (B, Sq, Sk, Dq, d_a, D)
        ctx.params = (Wq, bq, Wk, bk, Wv, bv, Wo, bo)
        none = torch.empty(0, device=dev)
        train = any(ctx.needs_input_grad)
        ctx.save_for_backward(Wq, Wk, Wv, Wo, qp.fh, xP.fh, op.hi if train else none, op.fh if train else none, lse, o.hi if train else none,
                              yP.hi if train else none)
        return out

    @staticmethod
    def backward(ctx, dout):
        Wq_, Wk_, Wv_, Wo_, qf, xf, oph, opf, lse, oh, yh = ctx.saved_tensors
        st, H, p = ctx.st, ctx.H, ctx.p
        qpack, kpack = ctx.packs
        B, Sq, Sk, Dq, d_a, D = ctx.dims
        Wq, bq, Wk, bk, Wv, bv, Wo, bo = ctx.params
        dk, Mq, Mk, Dr = D // H, B * Sq, B * Sk, H * d_a
        dev = dout.device
        dy2 = _f32c(dout).view(-1, Dq)
        has_res, res_p, res_site = ctx.res
        drop = None
        if has_res:
            dy2, drop = drop_grad(dy2, bo, res_p, res_site)
        P_, bias_done = grad_planes_from(dout, dy2, bo, drop, pack=qpack) if has_res else grad_planes(dy2, bo, drop=drop, pack=qpack)
        gbv = static_grad(bv)
        dbv_t = gbv if gbv is not None else (torch.zeros(D, device=dev, dtype=torch.float32) if bv is not None else None)
        do = linear_dx(P_, Wo, out_planes=Planes(torch.empty(Mq, D, device=dev, dtype=torch.bfloat16), None, Mq, D, pack=qpack), drop_post=True, drop_p=p,
                       site=ctx.site, colsum=dbv_t)
        if gbv is not None:
            grad_done(bv)
        dWo, dbo = wgrad(Wo, None if bias_done else bo, P_, Planes(oh, None, Mq, D, pack=qpack), dy2_for_bias=dy2)
        dop = Planes(torch.empty(Mq, Dr, device=dev, dtype=torch.bfloat16), None, Mq, Dr, pack=qpack)
        WvB = _value_planes(None, None, Wk, bk, Wv, bv, "bwd")
        gemm_bf16(do, Planes(WvB.hi, None, D, d_a), None, out_planes=dop, precision=PREC_BF16, b_km=True, a_blk=(d_a, dk), splitk=1)
        oP = Planes(oph, None, Mq, Dr, fh=opf, pack=qpack)
        dWv = _blockdiag_dw(do, oP, Wv, H)
        qp = Planes(None, None, Mq, Dr, fh=qf, pack=qpack)
        kP = Planes(None, None, Mk, d_a, fh=xf, pack=kpack)
        res = attn_bwd_planes(qp, kP, kP, oP, dop, lse, B, Sq, Sk, Dr, ctx.mask, H, 0.0, (st.c if bq is not None else None, None, None), fuse="kv",
                              kv_shared=True, scale=1.0 / math.sqrt(dk), bias_into=(_rank_dc(st), None, None), queue_bias=RANK_QUEUE_DC and _rank_deferred(st, Wq, bq, Wk))
        (Pq, dc), comb = res[0], res[3]
        dQ = dK = None
        if ctx.needs_input_grad[0]:      # dy = dq' W'
            dxq = torch.empty(Mq, Dq, device=dev, dtype=torch.float32)
            gemm_bf16(Pq, Planes(st.WpP.hi, None, Dr, Dq), dxq, ldc=Dq, precision=PREC_BF16, b_km=True)
            dQ = dxq.view(B, Sq, Dq)
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:      # dx = sum_h (dK'_h + dV'_h): [Mk][2 H d_a] against the identity stack
            dxk = torch.empty(Mk, d_a, device=dev, dtype=torch.float32)
            gemm_bf16(comb, Planes(st.Istack, None, 2 * Dr, d_a), dxk, ldc=d_a, precision=PREC_BF16, b_km=True)
            dK = dxk.view(B, Sk, d_a)    # (autograd adds dK and dV for the shared tensor; dV stays None)
        dWq, dbq, dWk = _rank_weight_grads(st, Pq, Planes(yh, None, Mq, Dq, pack=qpack), dc, Wq, bq, Wk)
        dbk = None
        if bk is not None:               # the key bias does not reach the output
            if static_grad(bk) is not None:
                grad_done(bk)
            else:
                dbk = torch.zeros_like(bk)
        return (dQ, dK, None, None, dWq, dbq, dWk, dbk, dWv, None if gbv is not None else dbv_t, dWo, dbo, None, None, None, None,
                (dout if has_res else None), None, None, None)


class FanoutFn(torch.autograd.Function):
    """x -> n aliases of x for n consumers: their gradients are added by library launches (bmt_add) in this node instead of by autograd's
    accumulation kernels (the encoder memories feed every decoder layer's key / value projections; a decoder layer's caption state feeds
    both encoder-decoder attentions)"""

    @staticmethod
    def forward(ctx, x, n):
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        gs = [g for g in gs if g is not None]
        if not gs:
            return None, None
        acc = gs[0]
        for i, g in enumerate(gs[1:]):
            a, b = _f32c(acc), _f32c(g)
            out = torch.empty_like(a)
            _lib.check(lib.bmt_add(_p(a), _p(b), _p(out), a.numel(), _st()), "bmt_add")
            acc = out
        return acc, None


def fanout(x, n: int):
    """n tensors that are x, for n consumers (planes attached to x travel with every alias)"""
    if n <= 1 or not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32 and x.requires_grad and torch.is_grad_enabled()):
        return (x,) * max(n, 1)
    outs = FanoutFn.apply(x, n)
    carry_pack(pack_of(x), *outs)
    pl = getattr(x, "_bmt_planes", None)
    if pl is not None and getattr(x, "_bmt_planes_version", x._version) == x._version:
        for o in outs:
            attach_planes(o, pl)
    return outs


# ----------------------------------------------------------------------------- step protocol (training_loop's bookkeeping as library launches)
def caption_shift(caption_idx: torch.Tensor, pad_idx: int):
    """(x, y, n_tokens) = (caption_idx[:, :-1], caption_idx[:, 1:], (y != pad_idx).sum()) of epoch_loops/captioning_epoch_loops.py:130-134 in
    ONE launch: x, y contiguous int64 (B, Tc), n_tokens an int64 0-dim tensor.  (Sliced views went through three ``.contiguous()`` copies --
    masks, embedding, loss -- and the count through a compare + a reduction: five framework kernels at the head of the step.)"""
    if not caption_idx.is_cuda or caption_idx.dtype != torch.int64 or caption_idx.dim() != 2 or caption_idx.stride(1) != 1 or caption_idx.shape[1] < 2:
        x, y = caption_idx[:, :-1], caption_idx[:, 1:]
        return x, y, (y != pad_idx).sum()
    B, T1 = caption_idx.shape
    xy = torch.empty(2, B, T1 - 1, device=caption_idx.device, dtype=torch.int64)
    n = torch.empty(1, device=caption_idx.device, dtype=torch.int64)
    _lib.check(lib.bmt_caption_shift(_p(caption_idx), caption_idx.stride(0), B, T1, int(pad_idx), _p(xy[0]), _p(xy[1]), _p(n), _st()), "bmt_caption_shift")
    return xy[0], xy[1], n.view(())


def loss_finish(kl: torch.Tensor, n_tokens: torch.Tensor, grad_scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """loss = kl / n_tokens (captioning_epoch_loops.py:135) and, into ``grad_scale`` (fp32 [1], optional), 1 / n_tokens -- one launch"""
    if not kl.is_cuda or kl.dtype != torch.float32 or n_tokens.dtype != torch.int64 or kl.numel() != 1 or n_tokens.numel() != 1:
        if grad_scale is not None:
            grad_scale.copy_((1.0 / n_tokens.to(torch.float32)).reshape(1))
        return kl / n_tokens
    loss = torch.empty((), device=kl.device, dtype=torch.float32)
    _lib.check(lib.bmt_loss_finish(_p(kl), _p(n_tokens), _p(loss), _p(grad_scale), _st()), "bmt_loss_finish")
    return loss


# ----------------------------------------------------------------------------- masks (bit-exact, no autograd)
def mask_from_features(feat: torch.Tensor, pad) -> torch.Tensor:
    """(feat[:, :, 0] != pad).unsqueeze(1) computed straight from the (B,S,D) stack (no slice copy)."""
    if feat.dim() == 2:   # already a (B,S) channel-0 slice
        B, S = feat.shape
        bs, ld = feat.stride(0), feat.stride(1)
    else:
        B, S = feat.shape[0], feat.shape[1]
        bs, ld = feat.stride(0), feat.stride(1)
    if feat.dtype != torch.float32:
        feat = feat.float()
        bs, ld = feat.stride(0), feat.stride(1)
    out = torch.empty(B, 1, S, device=feat.device, dtype=torch.uint8)
    _lib.check(lib.bmt_mask_from_features(_p(feat), bs, ld, float(pad), _p(out), B, S, _st()), "bmt_mask_from_features")
    return out.view(torch.bool)


def mask_from_tokens(trg: torch.Tensor, pad_idx: int, want_src: bool = True, want_trg: bool = True):
    t = trg.contiguous().long()
    B, S = t.shape
    src = torch.empty(B, 1, S, device=t.device, dtype=torch.uint8) if want_src else None
    tm = torch.empty(B, S, S, device=t.device, dtype=torch.uint8) if want_trg else None
    _lib.check(lib.bmt_mask_from_tokens(_p(t), int(pad_idx), _p(src), _p(tm), B, S, _st()), "bmt_mask_from_tokens")
    return (None if src is None else src.view(torch.bool)), (None if tm is None else tm.view(torch.bool))
