"""ctypes binding of libbmt_hip.so (the C ABI declared in include/bmt_hip.h).

The library is built in-tree by ``bmt_amd/csrc/build.sh`` (``__graft_entry__.build()``).
There is NO fallback: if the shared object is missing or a symbol is absent, importing the
ops raises -- the product path never silently degrades to eager PyTorch or to the oracle.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BMT_LIB_PATH") or os.path.join(_HERE, "lib", "libbmt_hip.so")   # override: A/B builds only

PREC_BF16, PREC_BF16X3, PREC_F16, PREC_F16W2 = 1, 3, 4, 5
EPI_BIAS, EPI_RELU, EPI_DROP_PRE, EPI_DROP_POST, EPI_RESIDUAL, EPI_GATE, EPI_ACCUM = 1, 2, 4, 8, 16, 32, 64

vp, i32, i64, f32, u32, u64 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint32, C.c_uint64


class GemmBf16Args(C.Structure):
    _fields_ = [("A_hi", vp), ("A_lo", vp), ("lda", i64), ("B_hi", vp), ("B_lo", vp), ("ldb", i64),
                ("C", vp), ("ldc", i64), ("C_hi", vp), ("C_lo", vp), ("ldp", i64),
                ("M", i32), ("N", i32), ("Kpad", i32), ("alpha", f32), ("flags", C.c_uint),
                ("bias", vp), ("residual", vp), ("ldr", i64), ("gate", vp), ("ldg", i64), ("gate_scale", f32),
                ("drop_p", f32), ("rng", vp), ("site", u32), ("precision", i32), ("splitk", i32),
                ("splitk_ws", vp), ("splitk_ws_bytes", i64), ("a_kmajor", i32), ("b_kmajor", i32), ("K", i32),
                ("conv_mode", i32), ("conv_cin", i32), ("conv_rows", i32), ("conv_S", i32), ("conv_halo", i32), ("colsum", vp), ("C_f16", vp),
                ("rows_dev", vp), ("c_row_dev", vp), ("m_dev", vp), ("a_blk_n", i32), ("a_blk_k", i32)]


class GemmBatch(C.Structure):
    _fields_ = [("nb_outer", i32), ("nb_inner", i32), ("a_off_o", i64), ("a_off_i", i64), ("b_off_o", i64), ("b_off_i", i64),
                ("b_rows_dev", vp), ("c_off_o", i64), ("c_off_i", i64), ("p_off_o", i64), ("p_off_i", i64), ("p2_off_o", i64), ("p2_off_i", i64),
                ("ldp2", i64), ("bias_off_i", i64), ("drop_off_o", i64), ("drop_off_i", i64),
                ("a_div", i32), ("c_div", i32), ("p_div", i32), ("p2_div", i32), ("a_qs", i64), ("c_qs", i64), ("p_qs", i64), ("p2_qs", i64)]


class AttnFwdArgs(C.Structure):
    _fields_ = [("Q", vp), ("K", vp), ("V", vp), ("O", vp), ("lse", vp),
                ("ldq", i64), ("ldk", i64), ("ldv", i64), ("ldo", i64),
                ("bsq", i64), ("bsk", i64), ("bsv", i64), ("bso", i64),
                ("mask", vp), ("mask_bs", i64), ("mask_qs", i64),
                ("B", i32), ("H", i32), ("Sq", i32), ("Sk", i32), ("dk", i32),
                ("scale", f32), ("drop_p", f32), ("rng", vp), ("site", u32), ("precision", i32)]


class AttnBwdArgs(C.Structure):
    _fields_ = [("Q", vp), ("K", vp), ("V", vp), ("O", vp), ("dO", vp), ("lse", vp),
                ("dQ", vp), ("dK", vp), ("dV", vp), ("delta_ws", vp),
                ("ldq", i64), ("ldk", i64), ("ldv", i64), ("ldo", i64),
                ("bsq", i64), ("bsk", i64), ("bsv", i64), ("bso", i64),
                ("mask", vp), ("mask_bs", i64), ("mask_qs", i64),
                ("B", i32), ("H", i32), ("Sq", i32), ("Sk", i32), ("dk", i32),
                ("scale", f32), ("drop_p", f32)]


class AttnFwdBf16Args(C.Structure):
    _fields_ = [("Qh", vp), ("Ql", vp), ("Kh", vp), ("Kl", vp), ("Vh", vp), ("Vl", vp), ("O", vp), ("lse", vp),
                ("ldq", i64), ("ldk", i64), ("ldv", i64), ("ldo", i64),
                ("bsq", i64), ("bsk", i64), ("bsv", i64), ("bso", i64),
                ("mask", vp), ("mask_bs", i64), ("mask_qs", i64),
                ("B", i32), ("H", i32), ("Sq", i32), ("Sk", i32), ("dk", i32),
                ("scale", f32), ("drop_p", f32), ("rng", vp), ("site", u32), ("precision", i32),
                ("Oh", vp), ("Ol", vp), ("ldop", i64), ("bsop", i64), ("Of", vp), ("q_off", vp), ("k_off", vp), ("b_order", vp), ("kv_shared", i32)]


class AttnBwdBf16Args(C.Structure):
    _fields_ = [("Qh", vp), ("Kh", vp), ("Vh", vp), ("O", vp), ("dO", vp), ("lse", vp),
                ("dQ", vp), ("dK", vp), ("dV", vp), ("delta_ws", vp), ("dOh_ws", vp),
                ("ldq", i64), ("ldk", i64), ("ldv", i64), ("ldo", i64),
                ("bsq", i64), ("bsk", i64), ("bsv", i64), ("bso", i64), ("dkv_ld", i64), ("dkv_bs", i64),
                ("mask", vp), ("mask_bs", i64), ("mask_qs", i64),
                ("B", i32), ("H", i32), ("Sq", i32), ("Sk", i32), ("dk", i32),
                ("scale", f32), ("drop_p", f32),
                ("Oh", vp), ("Ol", vp), ("ldop", i64), ("bsop", i64),
                ("dQh", vp), ("dKh", vp), ("dVh", vp), ("gq_ld", i64), ("gq_bs", i64), ("gkv_ld", i64), ("gkv_bs", i64),
                ("dQT", vp), ("dKT", vp), ("dVT", vp), ("gqT_ld", i64), ("gkvT_ld", i64),
                ("dbq", vp), ("dbk", vp), ("dbv", vp), ("Of", vp), ("kmean", vp), ("qkv_f16", i32),
                ("P_ws", vp), ("dS_ws", vp), ("Qb_ws", vp), ("bias_ws", vp), ("defer_bias", i32), ("q_off", vp), ("k_off", vp),
                ("rc_ws", vp), ("b_order", vp), ("kv_shared", i32)]


class CopyItem(C.Structure):
    _fields_ = [("src", vp), ("dst", vp), ("n", i64)]


class ColsumItem(C.Structure):
    _fields_ = [("part", vp), ("out", vp), ("rows", i32), ("D", i32), ("ld", i64)]


class SelectProposalsArgs(C.Structure):
    _fields_ = [("preds", vp), ("B", i32), ("S", i64), ("k", i32), ("flags", C.c_uint), ("durations", vp),
                ("min_len", f32), ("nms_thresh", f32), ("out", vp), ("out_idx", vp), ("count", vp), ("ws", vp),
                ("ws_bytes", C.c_size_t)]


PP_CORNERS, PP_TRIM, PP_FILTER = 1, 2, 4

# name -> (restype, argtypes); every symbol include/bmt_hip.h declares
SIGNATURES = {
    "bmt_version": (i32, []),
    "bmt_attn_kmean": (i32, [vp, i64, i64, vp, i64, i64, i32, i32, i32, vp, i32, vp, vp]),
    "bmt_gemm_bf16_grouped_ws_bytes": (C.c_size_t, [i32]),
    "bmt_gemm_small_outputs": (C.c_longlong, []),
    "bmt_gemm_small_batched": (i32, [C.POINTER(GemmBf16Args), C.POINTER(GemmBatch), vp]),
    "bmt_rank_prep": (i32, [vp, i64, i32, vp, i64, vp, i32, i32, i32, vp, vp, vp, i64, vp, vp, vp, vp]),
    "bmt_rank_chain": (i32, [vp, i64, i32, vp, i64, vp, i32, i32, i32, vp, vp, vp, i64, vp, i64, vp, vp]),
    "bmt_memory_transposed": (i32, [vp, i64, vp, i32, i32, i32, vp, vp, vp, vp]),
    "bmt_raw_softmax_fwd": (i32, [vp, vp, i32, i32, i32, i32, f32, vp, vp, i64, i64, vp]),
    "bmt_raw_softmax_bwd": (i32, [vp, vp, vp, i32, i32, i32, i32, f32, vp, i64, i64, vp]),
    "bmt_raw_attn_ok": (i32, [i32, i32]),
    "bmt_raw_attn_edges_ok": (i32, [i32, i32, i32]),
    "bmt_raw_attn_fwd_edges_ok": (i32, [i32, i32, i32]),
    "bmt_raw_attn_fwd_proj_ok": (i32, [i32, i32, i32, i32]),
    "bmt_raw_attn_bwd_proj_ok": (i32, [i32, i32, i32, i32]),
    "bmt_raw_attn_bwd_proj": (i32, [vp, i64, i32, vp, i64, f32, vp, u32, vp, vp, i64, vp, i64, vp, i64, i64, vp, i64, vp, vp, vp, i32, i32, i32, i32, i32, i32, f32, vp,
                                     i64, i64, vp, i64, vp, i64, vp, i64, vp, vp]),
    "bmt_raw_attn_fwd_proj": (i32, [vp, vp, i64, i32, vp, vp, i64, vp, vp, i64, vp, vp, i64, vp, i64, i64, vp, i64, vp, vp, i32, i32, i32, i32, i32, i32, f32, vp, vp,
                                     i64, i64, vp, vp, i64, vp]),
    "bmt_raw_attn_fwd_edges": (i32, [vp, vp, i64, vp, vp, i64, vp, i64, i64, vp, i64, vp, vp, i32, i32, i32, i32, i32, i32, f32, vp, vp, i64, i64, vp, vp, i64, vp]),
    "bmt_raw_attn_bwd_edges": (i32, [vp, i64, vp, i64, vp, i64, i64, vp, i64, vp, vp, vp, i32, i32, i32, i32, i32, i32, f32, vp, i64, i64, vp, i64, vp, i64, vp, i64,
                                      vp, vp]),
    "bmt_raw_attn_fwd": (i32, [vp, i64, i64, i64, vp, i64, vp, vp, i32, i32, i32, i32, i32, f32, vp, vp, i64, i64, vp, vp, i64, vp]),
    "bmt_raw_attn_bwd": (i32, [vp, i64, i64, i64, vp, i64, vp, vp, vp, i32, i32, i32, i32, i32, f32, vp, i64, i64, vp, i64, vp]),
    "bmt_gemm_bf16_grouped": (i32, [vp, i32, vp, C.c_size_t, vp]),
    "bmt_gemm_bf16_grouped_tables": (i32, [vp, i32, vp, C.c_size_t, C.POINTER(i32), vp]),
    "bmt_gemm_bf16_grouped_run": (i32, [vp, i32, C.POINTER(i32), vp]),
    "bmt_gemm_bf16_grouped_image": (i32, [vp, i32, vp, C.c_size_t, C.POINTER(i32)]),
    "bmt_planes_dropout": (i32, [vp, i64, i32, i32, vp, vp, vp, vp, i64, vp, vp, i64, vp, f32, vp, u32, vp, vp]),
    "bmt_layernorm_fwd_planes": (i32, [vp, i64, vp, vp, vp, i64, vp, vp, vp, vp, i32, i64, i32, i32, f32, vp, vp]),
    "bmt_layernorm_bwd_add": (i32, [vp, i64, vp, i64, vp, vp, vp, vp, i64, vp, i64, vp, vp, vp, i32, i32, vp, vp]),
    "bmt_npy_shape": (i32, [C.c_char_p, vp, vp, vp]),
    "bmt_npy_read_rows": (i32, [C.c_char_p, i64, i64, vp, i64, vp, vp]),
    "bmt_pad_batch": (i32, [vp, vp, i32, i32, i32, f32, vp, vp]),
    "bmt_select_proposals_ws_bytes": (C.c_size_t, [i32, i64, i32]),
    "bmt_select_proposals": (i32, [C.POINTER(SelectProposalsArgs), vp]),
    "bmt_transform_proposals": (i32, [vp, i32, i64, C.c_uint, vp, vp]),
    "bmt_last_error": (C.c_char_p, []),
    "bmt_device_cus": (i32, []),
    "bmt_gemm_bf16": (i32, [C.POINTER(GemmBf16Args), vp]),
    "bmt_pad_planes": (i32, [vp, i32, i32, i32, i32, i32, vp, vp, i32, i64, vp]),
    "bmt_planes": (i32, [vp, i64, i32, i32, vp, vp, vp, vp, i64, vp, vp, i64, vp, vp, vp]),
    "bmt_planes_desc_bytes": (i32, []),
    "bmt_planes_desc": (i32, [vp, vp, i64, i32, i32, vp, vp, vp, vp, i64, vp, vp, i64]),
    "bmt_planes_multi": (i32, [vp, i32, vp]),
    "bmt_planes_desc_tiles": (i32, [vp]),
    "bmt_planes_multi_flat": (i32, [vp, vp, i32, i32, vp]),
    "bmt_transpose_bf16": (i32, [vp, i64, i32, i32, vp, i64, vp]),
    "bmt_colsum": (i32, [vp, i64, i32, i32, vp, i32, vp, vp]),
    "bmt_colsum_multi": (i32, [vp, i32, vp]),
    "bmt_copy_multi": (i32, [vp, i32, vp]),
    "bmt_layernorm_bwd_partial": (i32, [vp, i64, vp, i64, vp, vp, vp, vp, i64, vp, i64, vp, i32, i32, vp, vp]),
    "bmt_attn_fwd": (i32, [C.POINTER(AttnFwdArgs), vp]),
    "bmt_attn_bwd": (i32, [C.POINTER(AttnBwdArgs), vp]),
    "bmt_attn_fwd_bf16": (i32, [C.POINTER(AttnFwdBf16Args), vp]),
    "bmt_attn_bwd_bf16": (i32, [C.POINTER(AttnBwdBf16Args), vp]),
    "bmt_attn_bwd_split_ws": (i32, [i32, i32, i32, i32, i32, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)]),
    "bmt_attn_bwd_rc_ws": (i32, [i32, i32, i32, i32, i32, C.POINTER(i64), C.POINTER(i64)]),
    "bmt_attn_bwd_bias_ws": (i64, [i32, i32, i32, i32, i32]),
    "bmt_layernorm_fwd": (i32, [vp, i64, vp, vp, vp, i64, vp, vp, i32, i32, f32, vp]),
    "bmt_layernorm_bwd_blocks": (i32, [i32]),
    "bmt_layernorm_bwd": (i32, [vp, i64, vp, i64, vp, vp, vp, vp, i64, i32, vp, vp, vp, i32, i32, vp]),
    "bmt_prep_features": (i32, [vp, vp, vp, vp, i32, i32, i32, f32, vp, u32, vp]),
    "bmt_pack_rows": (i32, [vp, i64, i32, i32, vp, vp, vp]),
    "bmt_pack_rows_ordered": (i32, [vp, i64, i32, i32, vp, vp, vp, vp]),
    "bmt_prep_features_packed": (i32, [vp, vp, vp, vp, i32, i32, i32, f32, vp, u32, vp, vp, vp]),
    "bmt_prep_embed": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, f32, f32, vp, u32, vp]),
    "bmt_prep_embed_bwd": (i32, [vp, vp, vp, i32, i32, i32, i32, f32, f32, vp, u32, vp]),
    "bmt_mask_from_features": (i32, [vp, i64, i64, f32, vp, i32, i32, vp]),
    "bmt_mask_from_tokens": (i32, [vp, i64, vp, vp, i32, i32, vp]),
    "bmt_dropout": (i32, [vp, vp, i64, f32, vp, u32, vp]),
    "bmt_gate": (i32, [vp, vp, f32, vp, i64, vp]),
    "bmt_dropout_add": (i32, [vp, vp, vp, i64, f32, vp, u32, vp]),
    "bmt_add": (i32, [vp, vp, vp, i64, vp]),
    "bmt_rng_advance": (i32, [vp, vp]),
    "bmt_copy3d": (i32, [vp, i64, i64, i64, vp, i32, i32, i32, i32, vp]),
    "bmt_log_softmax_fwd": (i32, [vp, i64, i32, i32, vp]),
    "bmt_log_softmax_bwd": (i32, [vp, i64, vp, i64, vp, i64, i32, i32, vp]),
    "bmt_ls_kl_fwd": (i32, [vp, i64, vp, vp, vp, i32, i32, f32, i64, vp]),
    "bmt_ls_kl_bwd": (i32, [vp, vp, i64, vp, vp, i32, i32, f32, i64, vp]),
    "bmt_log_softmax_fwd_stats": (i32, [vp, i64, i32, i32, vp, vp]),
    "bmt_ls_kl_fwd_stats": (i32, [vp, i64, vp, vp, vp, vp, i32, i32, f32, i64, vp]),
    "bmt_gen_lskl_bwd": (i32, [vp, i64, vp, vp, vp, i32, i32, f32, i64, vp, i64, vp, vp]),
    "bmt_planes_gate": (i32, [vp, i64, i32, i32, vp, vp, vp, vp, i64, vp, vp, i64, f32, vp]),
    "bmt_pad_planes_gate": (i32, [vp, vp, f32, i32, i32, i32, i32, i32, vp, i64, vp, vp]),
    "bmt_conv_weight_planes": (i32, [vp, i32, i32, i32, i32, vp, vp, vp, vp, i64, vp]),
    "bmt_conv_weight_grad": (i32, [vp, i64, i32, i32, i32, i32, vp, vp]),
    "bmt_zero": (i32, [vp, i64, vp]),
    "bmt_copy_h2d_async": (i32, [vp, vp, i64, vp]),
    "bmt_cat2": (i32, [vp, i64, i32, vp, i64, i32, vp, i64, i32, vp]),
    "bmt_split2": (i32, [vp, i64, vp, i64, i32, vp, i64, i32, i32, vp]),
    "bmt_caption_shift": (i32, [vp, i64, i32, i32, i64, vp, vp, vp, vp]),
    "bmt_loss_finish": (i32, [vp, vp, vp, vp, vp]),
    "bmt_layernorm_bwd_partial2": (i32, [vp, i64, vp, i64, vp, vp, vp, vp, i64, vp, i64, vp, i64, vp, i32, i32, vp, vp]),
    "bmt_layernorm_bwd_emit": (i32, [vp, i64, vp, i64, vp, vp, vp, vp, i64, vp, i64, vp, i64, vp, vp, i64, f32, vp, u32, i32, i32, vp, vp]),
    "bmt_adam_step": (i32, [vp, vp, i32, i64, vp, f32, f32, f32, f32, f32, vp, vp]),
    "bmt_grad_sqnorm": (i32, [vp, vp, i32, i64, vp, f32, vp, vp]),
    "bmt_scale_tensors": (i32, [vp, vp, i32, i64, vp, vp]),
    "bmt_targets_init": (i32, [vp, vp, vp, vp, i64, vp]),
    "bmt_make_targets": (i32, [vp, i32, vp, i32, i32, i32, f32, vp, vp, vp, vp, vp]),
    "bmt_prop_decode_loss": (i32, [vp, vp, i32, i32, i32, f32, vp, vp, vp, vp, vp, vp, vp]),
    "bmt_prop_decode_loss2": (i32, [vp, vp, i32, i32, i32, f32, vp, vp, vp, vp, vp, i64, vp, i32, vp]),
    "bmt_prop_loss_finalize_multi": (i32, [vp, i32, i32, vp, vp, f32, f32, vp, vp, vp]),
    "bmt_prop_loss_finalize": (i32, [vp, f32, f32, vp, vp]),
    "bmt_prop_loss_bwd": (i32, [vp, i32, i32, i32, vp, vp, vp, vp, vp, f32, f32, vp, vp, vp]),
}

_lib = None


def load():
    """Load (once) and type the shared library.  Raises if it is absent or incomplete."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with bmt_amd/csrc/build.sh (or __graft_entry__.build()). "
            "bmt_amd has no CPU / eager fallback by design.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.bmt_version() != 12:
        raise ImportError(f"libbmt_hip.so ABI version {lib.bmt_version()} != 12")
    _lib = lib
    return lib


ENOENT = -3     # BMT_ENOENT: a feature file cannot be opened
EALIGN = -4     # BMT_EALIGN: pointer / stride alignment requirement violated


def check(rc: int, what: str):
    if rc != 0:
        msg = load().bmt_last_error().decode(errors="replace")
        raise RuntimeError(f"{what} failed (rc={rc}): {msg}")
