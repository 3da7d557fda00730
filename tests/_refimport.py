"""Import the reference's L1 modules from /root/reference (build container only).

The reference is imported as-is (never copied): sys.path injection plus empty stub
modules for the third-party imports that are absent here and are not touched by the hot
path (spacy, pycocoevalcap.*, tensorboard) -- SURVEY.md 8(c).  Used by
tests/golden/make_golden.py and tests/test_oracle_vs_reference.py; both are skipped on
the GPU box, where /root/reference does not exist.
"""
import os
import sys
import types

REF = "/root/reference"


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REF, "model"))


def import_reference():
    """Returns a namespace of reference modules.  Idempotent."""
    if not reference_available():
        raise RuntimeError("reference not present")
    sys.dont_write_bytecode = True
    if REF not in sys.path:
        sys.path.insert(0, REF)

    def stub(name, **attrs):
        if name not in sys.modules:
            m = types.ModuleType(name)
            for k, v in attrs.items():
                setattr(m, k, v)
            sys.modules[name] = m
        return sys.modules[name]

    class _Dummy:  # never instantiated on the hot path
        def __init__(self, *a, **k):
            raise RuntimeError("stubbed third-party class")

    stub("spacy")
    stub("pycocoevalcap")
    stub("pycocoevalcap.tokenizer")
    stub("pycocoevalcap.tokenizer.ptbtokenizer", PTBTokenizer=_Dummy)
    stub("pycocoevalcap.bleu")
    stub("pycocoevalcap.bleu.bleu", Bleu=_Dummy)
    stub("pycocoevalcap.meteor")
    stub("pycocoevalcap.meteor.meteor", Meteor=_Dummy)
    stub("pycocoevalcap.rouge")
    stub("pycocoevalcap.rouge.rouge", Rouge=_Dummy)
    stub("pycocoevalcap.cider")
    stub("pycocoevalcap.cider.cider", Cider=_Dummy)

    # our own package also exposes `model` / `loss` sub-packages under bmt_amd.*, never at top level,
    # so the bare names below resolve to the reference.
    import importlib
    ns = types.SimpleNamespace()
    ns.multihead_attention = importlib.import_module("model.multihead_attention")
    ns.blocks = importlib.import_module("model.blocks")
    ns.encoders = importlib.import_module("model.encoders")
    ns.decoders = importlib.import_module("model.decoders")
    ns.generators = importlib.import_module("model.generators")
    ns.masking = importlib.import_module("model.masking")
    ns.captioning_module = importlib.import_module("model.captioning_module")
    ns.proposal_generator = importlib.import_module("model.proposal_generator")
    ns.proposal_utils = importlib.import_module("utilities.proposal_utils")
    ns.cap_loops = importlib.import_module("epoch_loops.captioning_epoch_loops")
    # loss/ has no __init__.py in the reference (the file is literally named "__init__py"):
    # namespace-package import works on py3.
    ns.label_smoothing = importlib.import_module("loss.label_smoothing")
    return ns
