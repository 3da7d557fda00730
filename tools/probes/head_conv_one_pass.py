#!/usr/bin/env python
"""GPU study: does a proposal head's k-tap Conv1d hold the 1e-3 bar of its outputs in ONE fp16 pass -- as it is, and
with a power-of-two scale on the first layer's weights (W0 2^s, b0 2^s, W1 2^-s: the same function in exact arithmetic; in fp16 the scaled
weights leave the subnormal range)?  The heads at the reference's real sizes against what the reference computed (tests/golden/prop_heads_real.npz).
Prints max |dy|, the worst of |dy| / (atol + rtol |y|) (the test's bar: < 1) and the time of the forward."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bmt_amd import ops  # noqa: E402
from bmt_amd.model.proposal_generator import ProposalGenerationHead  # noqa: E402

DEV = "cuda:0"


def main():
    g = np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "prop_heads_real.npz"), allow_pickle=True)
    for name in ("audio", "video"):
        d_in, k, T, anchors, seed = (int(v) for v in g[f"{name}/meta"])
        rows = torch.from_numpy(g[f"{name}/rows"]).long().to(DEV)
        want = torch.from_numpy(g[f"{name}/y"]).to(DEV)
        for label, prec, shift in (("fp16 x (fp16 hi+lo), 2 passes", ops.PREC_F16W2, 0), ("fp16, 1 pass", ops.PREC_F16, 0),
                                   ("fp16, 1 pass, W0 x 2^6", ops.PREC_F16, 6), ("fp16, 1 pass, W0 x 2^10", ops.PREC_F16, 10),
                                   ("bf16 x 3", ops.PREC_BF16X3, 0)):
            torch.manual_seed(seed)
            head = ProposalGenerationHead([d_in, 512, 512, 3 * anchors], k, 0.1, False).to(DEV).eval()
            convs = [m for m in head.conv_layers if isinstance(m, torch.nn.Conv1d)]
            with torch.no_grad():
                s = float(2 ** shift)
                convs[0].weight.mul_(s); convs[0].bias.mul_(s); convs[1].weight.div_(s)
            w0 = convs[0].weight.detach().abs()
            sub = float((w0[w0 > 0] < 6.1e-5).float().mean())
            old = ops.POLICIES["head_conv"].gemm
            ops.POLICIES["head_conv"].gemm = prec
            try:
                gen = torch.Generator().manual_seed(seed)
                x = (torch.randn(1, T, d_in, generator=gen).abs() * 0.25).to(DEV)
                with torch.no_grad():
                    y = head(x)
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(5):
                        head(x)
                    e1.record()
                    torch.cuda.synchronize()
            finally:
                ops.POLICIES["head_conv"].gemm = old
            got = y[0, rows]
            d = (got - want).abs()
            bar = float((d / (1e-3 + 1e-3 * want.abs())).max())
            print(f"{name:5s} k={k:3d} {label:32s} max|dy| {float(d.max()):.3e}  worst / bar {bar:5.2f}  |w0| < 2^-14: {sub:.2%}  forward {e0.elapsed_time(e1) / 5 * 1e3:7.1f} us", flush=True)


if __name__ == "__main__":
    main()
