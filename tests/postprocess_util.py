"""Seeded prediction tensors for the proposal post-processing tests (shared by the golden generator and the tests)."""
import torch

# tag -> (B videos, S candidates, k, seed, quantised confidences)
CASES = {
    "small": (3, 700, 100, 11, False),
    "ragged_k": (2, 37, 100, 12, False),        # fewer candidates than k
    "mid": (2, 50000, 100, 13, False),
    "ties": (2, 5000, 100, 14, True),
    "k1": (2, 300, 1, 15, False),
}


def make_preds(B, S, seed, ties=False):
    """(B, S, 3) [center_s, length_s, confidence] shaped like the generator's output: centres spread over and beyond the
    video, skewed lengths (some below the 0.2 s prior, some longer than the video), confidences in (0, 1); durations."""
    g = torch.Generator().manual_seed(seed)
    dur = [float(30 + 25 * i) for i in range(B)]
    # only uniform draws and + - * /: bit-reproducible on every host (exp / sigmoid / randn go through SIMD math libraries
    # whose last bit depends on the CPU)
    c = (torch.rand(B, S, generator=g) * 1.3 - 0.15) * torch.tensor(dur).view(B, 1)
    u = torch.rand(B, S, generator=g)
    ln = 0.05 + 60.0 * u * u * u
    ln[:, ::7] *= 0.01
    conf = torch.rand(B, S, generator=g)
    if ties:
        conf = (conf * 64).floor() / 64
    else:
        # pairwise distinct confidences per video (the reference's order is then unique): a random permutation of a grid
        conf = torch.stack([(torch.randperm(S, generator=g).float() + 0.5) / S for _ in range(B)])
    return torch.stack([c, ln, conf], dim=-1).contiguous(), dur
