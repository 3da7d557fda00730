"""Synthetic configs and batches for the train_cap / train_prop hot path.

Restates the *conventions* of the reference's data layer (not its code), which never
travels to the GPU box:
  * pad_idx=1, start=2, end=3, <unk>=0            (datasets/proposal_dataset.py:18-19)
  * rgb / audio tails padded with float(pad_idx), flow tails with 0
                                                   (datasets/captioning_dataset.py:256-258)
  * captions (B, Tc+1): [start, w_1..w_n, end, pad...]
  * cfg attribute names of utilities/config_constructor.py:74-98

Everything is generated from a private ``torch.Generator`` so global RNG state
(used for weight init) is untouched and results are reproducible across hosts.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Dict, Optional

import torch

PAD_IDX, START_IDX, END_IDX, UNK_IDX = 1, 2, 3, 0

KERNEL_SIZES = {"audio": [5, 13, 23, 35, 51, 69, 91, 121, 161, 211],
                "video": [1, 5, 9, 13, 19, 25, 35, 45, 61, 79]}


class Config(SimpleNamespace):
    """Attribute bag with the field names the reference's Config produces."""


def make_cfg(d_model: int = 1024, H: int = 4, N: int = 2, d_aud: int = 128, d_vid: int = 1024,
             d_model_caps: int = 300, dout_p: float = 0.1, device: str = "cuda:0", **kw) -> Config:
    cfg = Config(
        procedure="train_cap", modality="audio_video", use_linear_embedder=False,
        d_vid=d_vid, d_aud=d_aud, d_model_video=d_vid, d_model_audio=d_aud,
        d_model_caps=d_model_caps, d_model=d_model, H=H, N=N, dout_p=dout_p,
        d_ff_video=4 * d_vid, d_ff_audio=4 * d_aud, d_ff_caps=4 * d_model_caps,
        unfreeze_word_emb=False, pretrained_prop_model_path=None, finetune_prop_encoder=False,
        pretrained_cap_model_path=None, finetune_cap_encoder=False, layer_norm=False,
        anchors_num_audio=48, anchors_num_video=128, conv_layers_audio=[512, 512],
        conv_layers_video=[512, 512], kernel_sizes={k: list(v) for k, v in KERNEL_SIZES.items()},
        strides={"audio": 0.96, "video": 64 / 25}, pad_feats_up_to={"audio": 800, "video": 300},
        obj_coeff=1, noobj_coeff=100, max_prop_per_vid=100, nms_tiou_thresh=None,
        smoothing=0.7, lr=5e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, grad_clip=None,
        B=32, max_len=30, device=device, device_ids=[0], optimizer="adam",
    )
    for k, v in kw.items():
        setattr(cfg, k, v)
    return cfg


# the configurations BASELINE.json names
def cfg_config0(**kw) -> Config:
    """configs[0]: 1 bi-modal encoder/decoder layer, d_model=128, 10-word vocab, CPU-runnable."""
    return make_cfg(d_model=128, H=4, N=1, **kw)


def cfg_config1(**kw) -> Config:
    """configs[1]: N=2, d_model=1024, H=4 (the metric's configuration)."""
    return make_cfg(d_model=1024, H=4, N=2, **kw)


def cfg_tiny(**kw) -> Config:
    """fixture-only: odd small widths so every stored tensor is a few KB."""
    return make_cfg(d_model=128, H=4, N=1, d_aud=24, d_vid=48, d_model_caps=20, **kw)


class FakeVocab:
    def __init__(self, vectors):
        self.vectors = vectors


class FakeTrainDataset:
    """Stand-in for ActivityNetCaptionsDataset: the two attributes the model reads
    (model/captioning_module.py:121,145)."""

    def __init__(self, trg_voc_size: int, vectors: Optional[torch.Tensor]):
        self.trg_voc_size = trg_voc_size
        self.train_vocab = FakeVocab(vectors)
        self.pad_idx, self.start_idx, self.end_idx = PAD_IDX, START_IDX, END_IDX


def make_glove(voc_size: int, emb_dim: int, seed: int = 4321) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    return torch.randn(voc_size, emb_dim, generator=g) * 0.4


def make_cap_batch(cfg: Config, B: int, Tv: int, Ta: int, Tc: int, voc_size: int, seed: int = 1234,
                   ragged: bool = True) -> Dict:
    """One train_cap batch: feature_stacks {'rgb','flow','audio'} (fp32) and captions (B,Tc+1) int64.

    Valid lengths Lv ~ U[Tv/2, Tv], La = round(Lv*Ta/Tv); sample 0 is full length.  Channel 0 of valid
    rows is kept != pad value so masks derive exactly from the padding."""
    g = torch.Generator().manual_seed(seed)
    rgb = torch.randn(B, Tv, cfg.d_vid, generator=g).abs() * 0.25
    flow = torch.randn(B, Tv, cfg.d_vid, generator=g).abs() * 0.25
    audio = torch.randn(B, Ta, cfg.d_aud, generator=g).abs() * 0.25
    if ragged:
        Lv = torch.randint(max(Tv // 2, 1), Tv + 1, (B,), generator=g)
        Lv[0] = Tv
    else:
        Lv = torch.full((B,), Tv)
    La = torch.clamp(torch.round(Lv.float() * Ta / Tv).long(), 1, Ta)
    for b in range(B):
        rgb[b, Lv[b]:] = float(PAD_IDX)
        flow[b, Lv[b]:] = 0.0
        audio[b, La[b]:] = float(PAD_IDX)
    # valid rows must not look like padding in channel 0
    rgb[:, :, 0] = torch.where((rgb[:, :, 0] == float(PAD_IDX)) & (torch.arange(Tv)[None] < Lv[:, None]),
                               torch.full_like(rgb[:, :, 0], 0.5), rgb[:, :, 0])
    audio[:, :, 0] = torch.where((audio[:, :, 0] == float(PAD_IDX)) & (torch.arange(Ta)[None] < La[:, None]),
                                 torch.full_like(audio[:, :, 0], 0.5), audio[:, :, 0])
    lo_w = min(4, voc_size - 1)
    n_max = max(Tc - 2, 1)
    n_min = min(max(Tc * 8 // 30, 1), n_max)
    caps = torch.full((B, Tc + 1), PAD_IDX, dtype=torch.long)
    for b in range(B):
        n = int(torch.randint(n_min, n_max + 1, (1,), generator=g))
        words = torch.randint(lo_w, voc_size, (n,), generator=g)
        caps[b, 0] = START_IDX
        caps[b, 1:1 + n] = words
        caps[b, 1 + n] = END_IDX
    return {"feature_stacks": {"rgb": rgb, "flow": flow, "audio": audio}, "captions": caps,
            "Lv": Lv, "La": La}


def make_anchors(k: int, lo: float = 1.0, hi: float = 200.0):
    """Deterministic stand-in for the k-means anchor set (anchors are INPUTS to the path;
    k-means itself is out of scope): k log-spaced lengths in seconds, sorted."""
    return [float(lo * (hi / lo) ** (i / max(k - 1, 1))) for i in range(k)]


def make_prop_batch(cfg: Config, B: int, Tv: int, Ta: int, seed: int = 1234, events_per_video: int = 3) -> Dict:
    """One train_prop batch: full-video features padded like datasets/load_features.py:37-43 and
    targets (n_events,4) f32 [batch_idx, center_s, length_s, meta_idx]
    (datasets/proposal_dataset.py:133-166)."""
    base = make_cap_batch(cfg, B, Tv, Ta, 4, 10, seed=seed)
    g = torch.Generator().manual_seed(seed + 77)
    rows = []
    for b in range(B):
        dur = float(base["Lv"][b]) * cfg.strides["video"]
        n_ev = events_per_video + (b % 2)
        for e in range(n_ev):
            center = float(torch.rand(1, generator=g)) * dur
            length = float(math.exp(float(torch.rand(1, generator=g)) * math.log(min(200.0, max(dur, 2.0)))))
            rows.append([float(b), center, length, float(len(rows))])
    base["targets"] = torch.tensor(rows, dtype=torch.float32)
    return base
