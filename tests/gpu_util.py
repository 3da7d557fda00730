"""Helpers for the -m gpu parity tests (reports that localise a wrong tile/lane mapping from one run)."""
import numpy as np
import torch


def bf16_round(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).to(torch.float32)


def report(got: torch.Tensor, want: torch.Tensor, name: str) -> str:
    g = got.detach().double().cpu()
    w = want.detach().double().cpu()
    err = (g - w).abs()
    nan_g = torch.isnan(g) & ~torch.isnan(w)
    err = torch.where(torch.isnan(err), torch.full_like(err, float("inf")) * nan_g + 0.0, err)
    flat = err.flatten()
    imax = int(flat.argmax())
    idx = np.unravel_index(imax, tuple(err.shape)) if err.dim() > 0 else ()
    scale = float(w.abs().max()) if w.numel() else 0.0
    lines = [f"[{name}] shape={tuple(g.shape)} max|err|={float(flat.max()):.3e} at {idx} got={float(g.flatten()[imax]):.6g} "
             f"want={float(w.flatten()[imax]):.6g} max|want|={scale:.3e} mean|err|={float(flat[torch.isfinite(flat)].mean()):.3e} "
             f"nan_in_got={int(nan_g.sum())}"]
    if err.dim() >= 2:
        e2 = err.reshape(-1, err.shape[-1])
        R, Cc = e2.shape
        rb, cb = max(1, (R + 7) // 8), max(1, (Cc + 7) // 8)
        rows = []
        for i in range(0, R, rb):
            rows.append(" ".join(f"{float(e2[i:i + rb, j:j + cb].max()):8.1e}" for j in range(0, Cc, cb)))
        lines.append("max-err map (8x8 blocks over the last two dims flattened):\n" + "\n".join(rows))
    return "\n".join(lines)


def assert_close(got, want, atol, rtol=0.0, name="tensor"):
    g = got.detach().double().cpu()
    w = want.detach().double().cpu()
    assert g.shape == w.shape, f"[{name}] shape {tuple(g.shape)} != {tuple(w.shape)}"
    same_nan = torch.isnan(g) == torch.isnan(w)
    ok = ((g - w).abs() <= atol + rtol * w.abs()) | (torch.isnan(g) & torch.isnan(w))
    if not bool(ok.all() and same_nan.all()):
        raise AssertionError(report(got, want, name) + f"\n  tolerance atol={atol} rtol={rtol}; bad={int((~ok).sum())}/{ok.numel()}")


def rel_err(got, want) -> float:
    g, w = got.detach().double().cpu(), want.detach().double().cpu()
    return float((g - w).norm() / (w.norm() + 1e-30))
