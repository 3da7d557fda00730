"""CPU: the C-ABI library loads and exports every symbol include/bmt_hip.h declares (no compute calls)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "bmt_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bmt_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_hot_path():
    syms = declared_symbols()
    for need in ("bmt_gemm_bf16", "bmt_gemm_bf16_grouped", "bmt_planes", "bmt_attn_fwd", "bmt_attn_bwd", "bmt_attn_fwd_bf16", "bmt_attn_bwd_bf16",
                 "bmt_layernorm_fwd", "bmt_layernorm_bwd", "bmt_ls_kl_fwd", "bmt_adam_step", "bmt_pad_planes", "bmt_make_targets",
                 "bmt_prop_decode_loss", "bmt_last_error", "bmt_version", "bmt_attn_kmean"):
        assert need in syms


def test_library_loads_and_exports_everything():
    from bmt_amd import _lib
    lib = _lib.load()
    assert lib.bmt_version() == 12
    for s in declared_symbols():
        assert hasattr(lib, s), f"{s} declared in include/bmt_hip.h but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature in bmt_amd/_lib.py"
    for s in _lib.SIGNATURES:
        assert s in declared_symbols(), f"{s} bound in _lib.py but not declared in the header"


def test_errors_are_reported_not_thrown():
    """argument validation happens before any device work, so it is testable without a GPU."""
    import ctypes as C
    from bmt_amd import _lib
    lib = _lib.load()
    a = _lib.GemmBf16Args()
    rc = lib.bmt_gemm_bf16(C.byref(a), None)
    assert rc == -1 and b"null pointer" in lib.bmt_last_error()
    f = _lib.AttnFwdBf16Args()
    rc = lib.bmt_attn_fwd_bf16(C.byref(f), None)
    assert rc == -1 and b"null pointer" in lib.bmt_last_error()
    rc = lib.bmt_log_softmax_fwd(None, 0, 1, 1, None)
    assert rc == -1


def test_fused_cross_attention_entry_points_validate_before_any_device_work():
    """ABI 12: the predicates are pure host arithmetic (LDS budget of a (sample, head) workgroup), the launches refuse null pointers and shapes the
    predicate rejects without touching a device"""
    from bmt_amd import _lib
    lib = _lib.load()
    # configs[1]'s two memories fit; a 1024-wide memory with 1024 keys does not (64 (dm + 8) + 128 (Skp + 4) + 32 768 bytes <= 160 KB)
    assert lib.bmt_raw_attn_ok(128, 832) == 1 and lib.bmt_raw_attn_ok(1024, 256) == 1
    assert lib.bmt_raw_attn_ok(1024, 1024) == 0 and lib.bmt_raw_attn_ok(100, 256) == 0 and lib.bmt_raw_attn_ok(128, 1088) == 0
    assert lib.bmt_raw_attn_edges_ok(1024, 256, 256) == 1 and lib.bmt_raw_attn_edges_ok(1024, 64, 256) == 0        # (do_h is parked in the score tile's area)
    assert lib.bmt_raw_attn_fwd_edges_ok(1024, 256, 256) == 1 and lib.bmt_raw_attn_fwd_edges_ok(1024, 256, 100) == 0
    assert lib.bmt_raw_attn_fwd_proj_ok(1024, 256, 256, 320) == 1 and lib.bmt_raw_attn_fwd_proj_ok(128, 832, 256, 320) == 1
    assert lib.bmt_raw_attn_fwd_proj_ok(1024, 256, 256, 300) == 0 and lib.bmt_raw_attn_bwd_proj_ok(1024, 256, 256, 320) == 1
    rc = lib.bmt_raw_attn_fwd(None, 0, 0, 0, None, 0, None, None, 1, 1, 1, 128, 64, 1.0, None, None, 0, 0, None, None, 0, None)
    assert rc == -1 and b"bmt_raw_attn_fwd" in lib.bmt_last_error()
    rc = lib.bmt_raw_attn_bwd(None, 0, 0, 0, None, 0, None, None, None, 1, 1, 1, 128, 64, 1.0, None, 0, 0, None, 0, None)
    assert rc == -1 and b"bmt_raw_attn_bwd" in lib.bmt_last_error()


def test_missing_library_fails_loudly(monkeypatch):
    from bmt_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libbmt_hip.so")
    with pytest.raises(ImportError, match="no CPU / eager fallback"):
        _lib.load()


def test_error_codes_are_distinct():
    """a caller must be able to tell the BMT_E* codes apart (BMT_ENOENT and BMT_EALIGN once shared -3)."""
    from bmt_amd import _lib
    text = open(os.path.join(ROOT, "include", "bmt_hip.h")).read()
    codes = dict(re.findall(r"#define (BMT_E[A-Z]+) \((-\d+)\)", text))
    assert {"BMT_EINVAL", "BMT_EHIP", "BMT_ENOENT", "BMT_EALIGN"} <= set(codes)
    assert len(set(codes.values())) == len(codes), codes
    assert _lib.ENOENT == int(codes["BMT_ENOENT"]) and _lib.EALIGN == int(codes["BMT_EALIGN"])


def test_attention_backward_split_workspace_query():
    """bmt_attn_bwd_split_ws: sizes for the problems the split form takes, BMT_EINVAL (and zeros) for the ones it leaves to the two-kernel form"""
    import ctypes as C
    from bmt_amd import _lib
    lib = _lib.load()
    n = [C.c_int64(-1) for _ in range(3)]
    assert lib.bmt_attn_bwd_split_ws(32, 4, 800, 800, 256, *(C.byref(x) for x in n)) == 0
    # (Qb_ws: the scaled copy of q + one int of live-query bits per (batch, head, 128-query tile), rounded up to 16 bytes)
    assert n[0].value == 32 * 4 * 7 * 800 * 128 and n[1].value == 32 * 800 * 1024 + 2 * 32 * 4 * 7 and n[2].value == (32 * 7 + 2 * 32 * 7) * 1024
    for bad in ((32, 4, 29, 800, 256), (2, 4, 800, 800, 64), (2, 4, 800, 9000, 256)):
        assert lib.bmt_attn_bwd_split_ws(*bad, *(C.byref(x) for x in n)) == -1 and [x.value for x in n] == [0, 0, 0]


def test_recompute_backward_workspace_sizes():
    """bmt_attn_bwd_rc_ws (ABI 9): one int of live-query bits + one float of max |dO| per (batch, head, 128-query tile), the per-tile bias
    partials of the three gradients; BMT_EINVAL (and zeros) for the problems the recompute form leaves to the other forms"""
    import ctypes as C
    from bmt_amd import _lib
    lib = _lib.load()
    n = [C.c_int64(-1), C.c_int64(-1)]
    assert lib.bmt_attn_bwd_rc_ws(32, 4, 800, 256, 256, *(C.byref(x) for x in n)) == 0
    assert n[0].value == 2 * 32 * 4 * 7 and n[1].value == (32 * 7 + 2 * 32 * 2) * 1024
    for bad in ((32, 4, 30, 800, 256), (2, 4, 800, 800, 64), (1, 1, 4096, 128, 128), (1, 1, 128, 16384, 128)):
        assert lib.bmt_attn_bwd_rc_ws(*bad, *(C.byref(x) for x in n)) == -1 and [x.value for x in n] == [0, 0]
