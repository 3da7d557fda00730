"""CPU: the environment switches the product path reads are few, named, and each has a -m gpu test of its non-default arm.

Rounds 1-4 left 74 ``BMT_*`` switches behind (A/B arms that had lost: 44 ``getenv`` sites in the kernels' host code alone); a switch nobody
tests is a wrong-answer path behind an environment variable (one rotted silently in round 4).  What remains is listed here with the test
that exercises the arm that is NOT the default."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# switch -> (what the non-default arm does, the -m gpu test that runs it against a reference)
ALLOWED = {
    "BMT_LIB_PATH": ("another build of libbmt_hip.so (bmt_amd/_lib.py)", "tests/test_abi.py::test_missing_library_fails_loudly"),
    "BMT_PACK_ROWS": ("0: every padded position is computed, as the reference does",
                      "tests/test_gpu_packed.py::test_model_on_packed_rows_against_the_oracle_and_the_padded_path"),
    "BMT_RAW_MEMORY": ("0: the decoder's cross-attentions project the encoder memories to keys and values, as the reference does",
                       "tests/test_gpu_raw_memory.py::test_model_with_and_without_projected_keys_and_values"),
    "BMT_ENC_STREAMS": ("1: the whole pass on one stream", "tests/test_gpu_model.py::test_two_compute_streams_change_nothing_but_the_schedule"),
    "BMT_ATTN_BWD_SPLIT": ("0: the two-kernel attention backward everywhere; emit: the split form that leaves P / dS in HBM workspaces",
                           "tests/test_gpu_kernels.py::test_attention_backward_split_form"),
    "BMT_NO_FUSE_RES": ("1: LayerNorm / dropout_add / add as separate kernels",
                        "tests/test_gpu_model.py::test_fused_residual_block_equals_the_separate_kernels"),
    "BMT_FUSE_GEN_LOSS": ("0: generator and loss as separate autograd nodes", "tests/test_gpu_round4.py::test_generator_and_loss_as_one_node"),
    "BMT_NO_KMEAN": ("1: no mean-key correction of dQ", "tests/test_gpu_kernels.py::test_attention_backward_keys_with_a_common_component"),
    "BMT_LN_EMIT": ("0: every upstream gradient through its own conversion pass",
                    "tests/test_gpu_round4.py::test_layernorm_backward_emits_the_next_consumers_gradient_plane"),
}
# names with the prefix that are not environment switches: C macros of the kernels / the ABI
MACRO = re.compile(r"BMT_(EPI|PREC|OK$|E[A-Z]+$|CHECK|ABI|HIP_H|COLSUM_MAX|PP_|[A-Z0-9]+_(FETCH|STORE|STEP|FRAG|DMA|BAR|WAIT|W$|LD|LDA|FRAGS)|"
                   r"X_|P_|H_|W_|B_|K128_|DQ64|DKV|F64|GLOAD|LSTORE|COMPUTE|DMA|STEP|VMWAIT|LN$|LNF|FWD$|LS$|FETCH|F16_|DQ_|DQ16_|FWD_)")


def _switches_read():
    found = {}
    files = glob.glob(os.path.join(ROOT, "bmt_amd", "**", "*.py"), recursive=True) + glob.glob(os.path.join(ROOT, "bmt_amd", "csrc", "*"))
    for f in files:
        if not os.path.isfile(f):
            continue
        text = open(f, errors="replace").read()
        for m in re.finditer(r"(?:getenv\(\s*\"|environ(?:\.get)?[\(\[]\s*[\"'])(BMT_[A-Z0-9_]+)", text):
            found.setdefault(m.group(1), set()).add(os.path.relpath(f, ROOT))
    return found


def test_the_product_reads_only_the_listed_switches():
    found = _switches_read()
    extra = {k: sorted(v) for k, v in found.items() if k not in ALLOWED}
    assert not extra, f"environment switches outside the allow-list of tests/test_env_switches.py: {extra}"
    assert len(ALLOWED) <= 20


def test_no_getenv_is_left_in_the_kernels_host_code():
    for f in glob.glob(os.path.join(ROOT, "bmt_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "bmt_amd", "csrc", "*.h")):
        assert "getenv" not in open(f).read(), f


def test_every_switch_names_a_test_that_exists():
    for name, (_, test) in ALLOWED.items():
        path, _, fn = test.partition("::")
        text = open(os.path.join(ROOT, path)).read()
        assert f"def {fn}(" in text, f"{name}: {test} does not exist"
